// pt_shaders.hpp — the compiled-in surface shaders and the one table that registers them.
//
// The reference scans shaders/surface_shaders/*.glsl, numbers the files 1, 2, ... in sorted path
// order, splices each file's `SurfaceShaderResult f(SurfaceShaderData)` into the compute shader and
// generates two if/else chains that call f for its index (reference src/ShaderCompiler.cpp:93-118,
// 131-185; the structs: reference shaders/raytrace.comp:256-279).  Here a shader is a struct of
// device functions and PT_SURFACE_SHADERS is the registry: the host's path -> index table
// (pt_shader_count / pt_shader_path / pt_shader_index) and every dispatch site of the kernel are
// generated from it.  Adding a shader = one struct below + one row in the table, kept in sorted path
// order so that the indices are the ones the reference's directory scan would assign.
//
// The kernel casts one ray per lane per pass, so a GLSL shader that loops over the lights and calls
// getShadowFactor in the loop (toon.glsl:14-39) is written as three pieces and resumed once per light:
//   begin(data, result)   (each piece returns the result it was handed, changed) the code before the light loop, or the whole shader if it has no such loop
//   light(data, result, lightDir, distance, intensity)
//                         the loop body after `if (shadowFactor < 0.5) continue;` for one light that
//                         is not shadowed; lightDir / distance are lightVec / length(lightVec) and
//                         length(lightVec) (the shadow ray's direction and range: the same expressions
//                         on the same operands); may only change result.color
//   end(data, result)     the code after the loop
//   kLightLoop            whether light() is to be called at all
// Between the pieces only result.color is kept (in the lane's lighting accumulator), so a shader with
// a light loop sets result.emissive and result.emissiveStrength in end().  result arrives as the
// caller initialises it (raytrace.comp:311-318): color 0, emissive 0, emissiveStrength of the material.
// Of SurfaceShaderResult only color, emissive and emissiveStrength are ever read by the callers
// (raytrace.comp:324, 407-411), so the struct here carries those.
#pragma once

#include "pt_math.hpp"

namespace ptdev {

struct SurfaceShaderData { // raytrace.comp:256-268
  V3 hitPoint, normal, viewDir, surfaceColor;
  float ambient, diffuse, specular, shininess, emissiveStrength, time;
  int materialIndex;
};
struct SurfaceShaderResult { // raytrace.comp:271-279, the fields the callers read
  V3 color, emissive;
  float emissiveStrength;
};

// ---- shaders/surface_shaders/rainbow.glsl ----
// rainbow.glsl:1-8
__device__ __forceinline__ float hue2rgb(float p, float q, float t) {
  if (t < 0.0f) t += 1.0f;
  if (t > 1.0f) t -= 1.0f;
  if (t < 1.0f / 6.0f) return p + ((q - p) * 6.0f) * t;
  if (t < 0.5f) return q;
  if (t < 2.0f / 3.0f) return p + ((q - p) * (2.0f / 3.0f - t)) * 6.0f;
  return p;
}
// rainbow.glsl:10-22
__device__ __forceinline__ V3 hsl2rgb(float h, float s, float l) {
  float H = glfract(h);
  float S = glclamp(s, 0.0f, 1.0f);
  float L = glclamp(l, 0.0f, 1.0f);
  if (S == 0.0f) return mk3(L, L, L);
  float q = (L < 0.5f) ? (L * (1.0f + S)) : ((L + S) - L * S);
  float p = 2.0f * L - q;
  return mk3(hue2rgb(p, q, H + 1.0f / 3.0f), hue2rgb(p, q, H), hue2rgb(p, q, H - 1.0f / 3.0f));
}
// rainbow.glsl:24-50 — result.emissive (result.color stays 0)
__device__ __attribute__((noinline)) V3 rainbow_emissive(V3 hitPoint, float emissiveStrength, float time) {
  V3 sweepDir = normalize3(mk3(1.0f, 0.5f, 0.25f));
  float phase = dot3(hitPoint, sweepDir) * 2.0f + time * 0.002f;
  float cosVal = cosf(phase);
  float hue = glfract(0.5f + 0.5f * cosVal);
  return scale3(hsl2rgb(hue, 1.0f, 0.5f), emissiveStrength);
}
struct RainbowShader {
  static constexpr bool kLightLoop = false;
  static __device__ __forceinline__ SurfaceShaderResult begin(const SurfaceShaderData &data, SurfaceShaderResult result) {
    result.emissive = rainbow_emissive(data.hitPoint, result.emissiveStrength, data.time);
    return result;
  }
  static __device__ __forceinline__ SurfaceShaderResult light(const SurfaceShaderData &, SurfaceShaderResult result, V3, float, float) { return result; }
  static __device__ __forceinline__ SurfaceShaderResult end(const SurfaceShaderData &, SurfaceShaderResult result) { return result; }
};

// ---- shaders/surface_shaders/toon.glsl ----
struct ToonShader {
  static constexpr bool kLightLoop = true;
  static __device__ __forceinline__ SurfaceShaderResult begin(const SurfaceShaderData &data, SurfaceShaderResult result) {
    result.color = scale3(data.surfaceColor, data.ambient); // toon.glsl:11
    return result;
  }
  static __device__ __forceinline__ SurfaceShaderResult light(const SurfaceShaderData &data, SurfaceShaderResult result, V3 lightDir,
                                                              float distance, float intensity) { // toon.glsl:22-39
    const float attenuation = intensity / ((1.0f + 0.09f * distance) + (0.032f * distance) * distance);
    const float diffuseRaw = glmax(0.0f, dot3(data.normal, lightDir));
    const float diffuseQuantized = diffuseRaw > 0.6f ? 1.0f : (diffuseRaw > 0.3f ? 0.5f : 0.0f);
    const V3 diffuse = scale3(scale3(data.surfaceColor, data.diffuse * diffuseQuantized), attenuation);
    const V3 halfDir = normalize3(add3(lightDir, data.viewDir));
    const float specularRaw = highlight_pow(glmax(0.0f, dot3(data.normal, halfDir)), data.shininess);
    const float specularQuantized = specularRaw > 0.8f ? 1.0f : 0.0f;
    const V3 specular = scale3(scale3(scale3(data.surfaceColor, data.specular), specularQuantized), attenuation);
    result.color = add3(result.color, add3(diffuse, specular));
    return result;
  }
  static __device__ __forceinline__ SurfaceShaderResult end(const SurfaceShaderData &data, SurfaceShaderResult result) { // toon.glsl:42-47
    if (data.emissiveStrength > 0.0f) result.emissive = scale3(data.surfaceColor, data.emissiveStrength);
    return result;
  }
};

// ---- the registry: X(index, path, shader struct), in sorted path order, indices 1, 2, ... ----
#define PT_SURFACE_SHADERS(X)                                 \
  X(1, "shaders/surface_shaders/rainbow.glsl", RainbowShader) \
  X(2, "shaders/surface_shaders/toon.glsl", ToonShader)

// Dispatch, generated from the registry (an if/else chain like the one the reference generates,
// src/ShaderCompiler.cpp:162-170).  An index no row has leaves `result` as it came, which is what the
// reference's chain does when no branch matches (color 0 + emissive 0).
__device__ __forceinline__ bool shader_has_light_loop(int index) {
#define PT_X(IDX, PATH, S) if (index == IDX) return S::kLightLoop;
  PT_SURFACE_SHADERS(PT_X)
#undef PT_X
  return false;
}
__device__ __forceinline__ SurfaceShaderResult shader_begin(int index, const SurfaceShaderData &data, SurfaceShaderResult result) {
#define PT_X(IDX, PATH, S) if (index == IDX) return S::begin(data, result);
  PT_SURFACE_SHADERS(PT_X)
#undef PT_X
  return result;
}
__device__ __forceinline__ SurfaceShaderResult shader_light(int index, const SurfaceShaderData &data, SurfaceShaderResult result,
                                                            V3 lightDir, float distance, float intensity) {
#define PT_X(IDX, PATH, S) if (S::kLightLoop && index == IDX) return S::light(data, result, lightDir, distance, intensity);
  PT_SURFACE_SHADERS(PT_X)
#undef PT_X
  return result;
}
__device__ __forceinline__ SurfaceShaderResult shader_end(int index, const SurfaceShaderData &data, SurfaceShaderResult result) {
#define PT_X(IDX, PATH, S) if (index == IDX) return S::end(data, result);
  PT_SURFACE_SHADERS(PT_X)
#undef PT_X
  return result;
}

} // namespace ptdev
