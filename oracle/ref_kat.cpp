// ref_kat.cpp — golden-vector generator driven by the REFERENCE's own classes.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile (target `ref`)
// against the reference's legacy double-precision sources where they lie
// (/root/reference/src/{Vec3,Ray,Quaternion,Camera,Object,Material,Sphere}.cpp,
// none of which need Vulkan/GLM/SDL); the binary goes to oracle/_ref/ and is
// never committed.  Its JSON output is committed as tests/golden/ref_kat.json.
//
// What the reference functions pin:
//   Sphere::hit              (reference src/Sphere.cpp:9-46)   the same quadratic as
//                            the shader's hitSphere (shaders/raytrace.comp:142-168)
//   Quaternion::rotateVector (reference src/Quaternion.cpp:178-186)  equals the
//                            shader's rotateByQuaternion for unit quaternions
//   Camera::getRay           (reference src/Camera.cpp:38-41)  spans twice the
//                            shader's field extent, so shader uv maps to
//                            u = 0.5 + (uv.x-0.5)/2, v = 0.5 + (uv.y-0.5)/2
//   Quaternion algebra       fromAxisAngle / operator* / slerp / toEulerAngles,
//                            used to check the host-side Quaternion class
#include "Camera.h"
#include "Quaternion.h"
#include "Ray.h"
#include "Sphere.h"
#include "Vec3.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>

static uint64_t g_state = 0x853c49e6748fea9bULL;
static uint32_t rnd32() { // splitmix64, high half
  uint64_t z = (g_state += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}
// uniform float32 in [lo, hi), returned as the exactly representable float
static float rndf(float lo, float hi) {
  float u = (float)(rnd32() >> 8) * (1.0f / 16777216.0f);
  return lo + (hi - lo) * u;
}

static void pv(const char *k, const Vec3 &v, bool comma = true) {
  std::printf("\"%s\": [%.17g, %.17g, %.17g]%s", k, v.x, v.y, v.z, comma ? ", " : "");
}

int main() {
  std::printf("{\n");

  // ---- Sphere::hit ----
  std::printf("\"sphere_hit\": [\n");
  const int NS = 256;
  for (int i = 0; i < NS; i++) {
    Vec3 o(rndf(-6, 6), rndf(-1, 4), rndf(-2, 8));
    Vec3 c(rndf(-4, 4), rndf(-1, 3), rndf(-6, 3));
    double r = rndf(0.15f, 1.5f);
    // aim near the sphere so about half of the rays hit
    Vec3 aim(c.x + rndf(-1.5f, 1.5f) * r, c.y + rndf(-1.5f, 1.5f) * r, c.z + rndf(-1.5f, 1.5f) * r);
    Vec3 dd = (aim - o).normalized();
    // round the direction to float32 and feed the rounded value to the reference
    Vec3 d((float)dd.x, (float)dd.y, (float)dd.z);
    if (i % 16 == 15) { // origin inside the sphere: far root
      o = Vec3((float)(c.x + 0.3 * r), (float)c.y, (float)c.z);
    }
    double tMin = 0.001f, tMax = (i % 8 == 7) ? rndf(0.5f, 6.0f) : 1e10f;
    Sphere s(c, r, std::shared_ptr<Material>());
    HitRecord rec{};
    bool hit = s.hit(Ray(o, d), tMin, tMax, rec);
    std::printf("{");
    pv("origin", o); pv("dir", d); pv("center", c);
    std::printf("\"radius\": %.17g, \"tmin\": %.17g, \"tmax\": %.17g, \"hit\": %d, ", r, tMin, tMax, hit ? 1 : 0);
    if (hit) {
      std::printf("\"t\": %.17g, ", rec.t);
      pv("point", rec.point); pv("normal", rec.normal, false);
    } else {
      std::printf("\"t\": 0");
    }
    std::printf("}%s\n", i + 1 < NS ? "," : "");
  }
  std::printf("],\n");

  // ---- Quaternion::rotateVector with float32-rounded unit quaternions ----
  std::printf("\"quat_rotate\": [\n");
  const int NQ = 128;
  for (int i = 0; i < NQ; i++) {
    Vec3 axis(rndf(-1, 1), rndf(-1, 1), rndf(-1, 1));
    double angle = rndf(-6.3f, 6.3f);
    Quaternion q = Quaternion::fromAxisAngle(axis, angle);
    Quaternion qf((float)q.w, (float)q.x, (float)q.y, (float)q.z);
    Vec3 v(rndf(-5, 5), rndf(-5, 5), rndf(-5, 5));
    Vec3 r = qf.rotateVector(v);
    Vec3 ri = qf.conjugate().rotateVector(v);
    std::printf("{\"q_xyzw\": [%.17g, %.17g, %.17g, %.17g], ", qf.x, qf.y, qf.z, qf.w);
    pv("v", v); pv("rotated", r); pv("inv_rotated", ri, false);
    std::printf("}%s\n", i + 1 < NQ ? "," : "");
  }
  std::printf("],\n");

  // ---- Camera::getRay at the shader's uv ----
  std::printf("\"camera_ray\": [\n");
  const int NC = 64;
  for (int i = 0; i < NC; i++) {
    int W = (i % 2) ? 1920 : 640, H = (i % 2) ? 1080 : 480;
    double th = i / 180.0 * 7.0;
    Vec3 eye((float)(2.0 + 3.0 * std::cos(th)), 1.5, (float)(6.0 + 3.0 * std::sin(th)));
    Camera cam(eye, Vec3(2, 1.5, 0), Vec3(0, 1, 0), 60.0, (double)W / H);
    int px = (int)(rnd32() % (uint32_t)W), py = (int)(rnd32() % (uint32_t)H);
    double uvx = (double)px / W, uvy = 1.0 - (double)py / H;
    Ray r = cam.getRay(0.5 + (uvx - 0.5) / 2.0, 0.5 + (uvy - 0.5) / 2.0);
    std::printf("{\"width\": %d, \"height\": %d, \"px\": %d, \"py\": %d, ", W, H, px, py);
    pv("eye", eye); pv("dir", r.direction, false);
    std::printf("}%s\n", i + 1 < NC ? "," : "");
  }
  std::printf("],\n");

  // ---- Quaternion algebra (for the host-side Quaternion class) ----
  std::printf("\"quat_algebra\": [\n");
  const int NA = 48;
  for (int i = 0; i < NA; i++) {
    Vec3 a1(rndf(-1, 1), rndf(-1, 1), rndf(-1, 1)), a2(rndf(-1, 1), rndf(-1, 1), rndf(-1, 1));
    double g1 = rndf(-3, 3), g2 = rndf(-3, 3), t = rndf(0, 1);
    Quaternion q1 = Quaternion::fromAxisAngle(a1, g1), q2 = Quaternion::fromAxisAngle(a2, g2);
    Quaternion m = q1 * q2, s = Quaternion::slerp(q1, q2, t), l = Quaternion::lerp(q1, q2, t);
    Quaternion e = Quaternion::fromEulerAngles(g1, g2 * 0.4, t);
    double roll, pitch, yaw;
    m.toEulerAngles(roll, pitch, yaw);
    Vec3 ax = q1.getAxis();
    std::printf("{");
    pv("axis1", a1); pv("axis2", a2);
    std::printf("\"angle1\": %.17g, \"angle2\": %.17g, \"t\": %.17g, ", g1, g2, t);
    std::printf("\"q1_wxyz\": [%.17g, %.17g, %.17g, %.17g], ", q1.w, q1.x, q1.y, q1.z);
    std::printf("\"q2_wxyz\": [%.17g, %.17g, %.17g, %.17g], ", q2.w, q2.x, q2.y, q2.z);
    std::printf("\"mul_wxyz\": [%.17g, %.17g, %.17g, %.17g], ", m.w, m.x, m.y, m.z);
    std::printf("\"slerp_wxyz\": [%.17g, %.17g, %.17g, %.17g], ", s.w, s.x, s.y, s.z);
    std::printf("\"lerp_wxyz\": [%.17g, %.17g, %.17g, %.17g], ", l.w, l.x, l.y, l.z);
    std::printf("\"euler_in\": [%.17g, %.17g, %.17g], ", g1, g2 * 0.4, t);
    std::printf("\"from_euler_wxyz\": [%.17g, %.17g, %.17g, %.17g], ", e.w, e.x, e.y, e.z);
    std::printf("\"mul_to_euler\": [%.17g, %.17g, %.17g], ", roll, pitch, yaw);
    std::printf("\"q1_angle\": %.17g, ", q1.getAngle());
    pv("q1_axis", ax, false);
    std::printf("}%s\n", i + 1 < NA ? "," : "");
  }
  std::printf("]\n}\n");
  return 0;
}
