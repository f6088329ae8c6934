// glm_compat.h — the handful of GLM types the scene API uses as plain data.
//
// The reference's scene wrappers store glm::vec3 / vec4 / ivec3 / mat4
// (reference include/SceneWrappers.h, include/Ellipsoid.h,
// include/VulkanRenderer.h:16-71).  When <glm/glm.hpp> is installed it is used;
// otherwise these layout-compatible stand-ins are, so code written against the
// reference's API compiles unchanged.
#ifndef PATH_TRACER_GLM_COMPAT_H
#define PATH_TRACER_GLM_COMPAT_H

#if defined(__has_include)
#if __has_include(<glm/glm.hpp>) && !defined(PATH_TRACER_NO_GLM)
#define PATH_TRACER_HAVE_GLM 1
#endif
#endif

#ifdef PATH_TRACER_HAVE_GLM
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#else
#include <cmath>

namespace glm {

struct vec3 {
  float x, y, z;
  constexpr vec3() noexcept : x(0), y(0), z(0) {}
  constexpr explicit vec3(float s) noexcept : x(s), y(s), z(s) {}
  constexpr vec3(float x_, float y_, float z_) noexcept : x(x_), y(y_), z(z_) {}
  constexpr float operator[](int i) const noexcept { return i == 0 ? x : (i == 1 ? y : z); }
};
constexpr vec3 operator+(vec3 a, vec3 b) noexcept { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
constexpr vec3 operator-(vec3 a, vec3 b) noexcept { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
constexpr vec3 operator*(vec3 a, float s) noexcept { return {a.x * s, a.y * s, a.z * s}; }
constexpr vec3 operator*(float s, vec3 a) noexcept { return a * s; }
constexpr float dot(vec3 a, vec3 b) noexcept { return a.x * b.x + a.y * b.y + a.z * b.z; }
constexpr vec3 cross(vec3 a, vec3 b) noexcept {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(vec3 a) noexcept { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) noexcept { return a * (1.0f / length(a)); }

struct ivec3 {
  int x, y, z;
  constexpr ivec3() noexcept : x(0), y(0), z(0) {}
  constexpr explicit ivec3(int s) noexcept : x(s), y(s), z(s) {}
  constexpr ivec3(int x_, int y_, int z_) noexcept : x(x_), y(y_), z(z_) {}
};

struct vec4 {
  float x, y, z, w;
  constexpr vec4() noexcept : x(0), y(0), z(0), w(0) {}
  constexpr explicit vec4(float s) noexcept : x(s), y(s), z(s), w(s) {}
  constexpr vec4(float x_, float y_, float z_, float w_) noexcept : x(x_), y(y_), z(z_), w(w_) {}
  constexpr vec4(vec3 v, float w_) noexcept : x(v.x), y(v.y), z(v.z), w(w_) {}
};

// Column-major 4x4, m[column][row] like GLM.
struct mat4 {
  float m[4][4];
  constexpr mat4() noexcept : m{{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}} {}
  constexpr explicit mat4(float d) noexcept : m{{d, 0, 0, 0}, {0, d, 0, 0}, {0, 0, d, 0}, {0, 0, 0, d}} {}
  float *operator[](int c) noexcept { return m[c]; }
  const float *operator[](int c) const noexcept { return m[c]; }
};
inline mat4 operator*(const mat4 &a, const mat4 &b) noexcept {
  mat4 r;
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      float s = 0.0f;
      for (int k = 0; k < 4; ++k) s += a.m[k][row] * b.m[c][k];
      r.m[c][row] = s;
    }
  return r;
}
constexpr float radians(float deg) noexcept { return deg * 0.017453292519943295f; }

// Right-handed look-at, as glm::lookAt.
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) noexcept {
  const vec3 f = normalize(center - eye);
  const vec3 s = normalize(cross(f, up));
  const vec3 u = cross(s, f);
  mat4 r(1.0f);
  r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
  r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
  r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
  r[3][0] = -dot(s, eye); r[3][1] = -dot(u, eye); r[3][2] = dot(f, eye);
  return r;
}
// Right-handed, depth -1..1, as glm::perspective's default.
inline mat4 perspective(float fovy, float aspect, float zNear, float zFar) noexcept {
  const float t = std::tan(fovy / 2.0f);
  mat4 r;
  r[0][0] = 1.0f / (aspect * t);
  r[1][1] = 1.0f / t;
  r[2][2] = -(zFar + zNear) / (zFar - zNear);
  r[2][3] = -1.0f;
  r[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
  return r;
}

} // namespace glm
#endif // PATH_TRACER_HAVE_GLM
#endif
