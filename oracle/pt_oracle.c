/*
 * pt_oracle.c — float32 CPU restatement of the reference's per-pixel ray loop
 * (reference shaders/raytrace.comp + shaders/surface_shaders/{rainbow,toon}.glsl).
 *
 * TEST INFRASTRUCTURE ONLY — see pt_oracle.h.  PARITY UNPINNED BY THE
 * REFERENCE (it has no tests or golden vectors for this path); pinned by the
 * known-answer tests and the legacy-class vectors under tests/golden/.
 *
 * Every function names the GLSL it follows.  Build with
 *   gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -mfma
 * so the compiler neither fuses nor reorders anything: the only fused
 * multiply-adds are the explicit fmaf() calls below.
 *
 * Arithmetic contract (GLSL leaves contraction and the precision of
 * normalize/pow/cos to the implementation; this file fixes one legal choice):
 *   dot(a,b)            = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   cross(a,b).x        = fma(a.y,b.z, -(a.z*b.y))      (y, z cyclic)
 *   a + s*b  (points)   = fma(s, b, a) per component    (madd3)
 *   length(v)           = sqrt(dot(v,v))
 *   normalize(v)        = v * (1 / length(v))
 *   vec / scalar        = per-component IEEE division
 *   hitSphere:     c    = fma(-r, r, dot(oc,oc));  disc = fma(hb, hb, -c)
 *   hitEllipsoid:  disc = fma(hb, hb, -(a*c))
 *   volume march:  acc  = fma(1-acc, alpha, acc)
 *   everything else     = one IEEE operation per GLSL operator, left to right
 *   pow, cos            = double-precision libm rounded to float
 *   rgba8 store         = round-half-even of c*255 after clamp; NaN -> 0
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static const float MAX_DISTANCE = 1e10f; /* raytrace.comp:94 */
static const float EPSILON = 0.001f;     /* raytrace.comp:95 */

/* ---------- vector helpers (the contract above) ---------- */
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 div3s(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) {
  return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
}
static inline v3 cross3(v3 a, v3 b) {
  return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)),
            fmaf(a.x, b.y, -(a.y * b.x)));
}
/* a + s*b */
static inline v3 madd3(v3 a, float s, v3 b) {
  return V3(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z));
}
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) {
  float inv = 1.0f / length3(a);
  return scale3(a, inv);
}
/* GLSL min/max/clamp: max(x,y) = (x<y)?y:x ; min(x,y) = (y<x)?y:x */
static inline float gl_max(float x, float y) { return (x < y) ? y : x; }
static inline float gl_min(float x, float y) { return (y < x) ? y : x; }
static inline float gl_clamp(float x, float lo, float hi) {
  return gl_min(gl_max(x, lo), hi);
}
static inline float gl_fract(float x) { return x - floorf(x); }
static inline float pow_f(float x, float y) { return (float)pow((double)x, (double)y); }
static inline float cos_f(float x) { return (float)cos((double)x); }
/* float -> int with the device's conversion semantics (saturating, NaN -> 0) */
static inline int f2i(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return (-2147483647 - 1);
  return (int)x;
}

typedef struct { v3 origin, direction; } Ray;
typedef struct { v3 point, normal; float t; int materialIndex; } HitRecord;

typedef struct {
  const PtoScene *s;
  const PtPushConstants *pc;
  int width, height, max_bounces;
  int volume_filter;             /* 0 nearest (the reference), 1 trilinear (extension) */
  int first_sample, num_samples; /* accumulation (extension); 0, 1 = the reference's single sample */
  uint64_t cnt[PT_CNT_COUNT];
} Ctx;

/* Per-thread pointer to the {sphere, ellipsoid} root-evaluation counters of
 * the frame being rendered (NULL inside the single-function probes). */
static _Thread_local uint64_t *g_roots = 0;

/* ---------- raytrace.comp:99-115 ---------- */
static v3 rotateByQuaternion(v3 v, v4 q) {
  v3 qxyz = V3(q.x, q.y, q.z);
  float qw = q.w;
  v3 t = scale3(cross3(qxyz, v), 2.0f);
  /* v + qw*t + cross(qxyz, t), evaluated (v + qw*t) + cross */
  return add3(madd3(v, qw, t), cross3(qxyz, t));
}
static v3 inverseRotateByQuaternion(v3 v, v4 q) {
  v4 qc = {-q.x, -q.y, -q.z, q.w};
  return rotateByQuaternion(v, qc);
}

/* ---------- raytrace.comp:121-140 ---------- */
static Ray createCameraRay(const Ctx *c, float uvx, float uvy) {
  Ray ray;
  ray.origin = ld3(c->pc->cameraPos);
  v3 target = V3(2.0f, 1.5f, 0.0f);
  v3 forward = normalize3(sub3(target, ray.origin));
  v3 right = normalize3(cross3(forward, V3(0.0f, 1.0f, 0.0f)));
  v3 up = normalize3(cross3(right, forward));
  /* radians(60.0) = 60 * pi/180 in float */
  float fov = tanf((60.0f * 0.017453292519943295f) / 2.0f);
  float aspectRatio = (float)c->width / (float)c->height;
  float sx = ((uvx - 0.5f) * aspectRatio) * fov;
  float sy = (uvy - 0.5f) * fov;
  ray.direction = normalize3(madd3(madd3(forward, sx, right), sy, up));
  return ray;
}

/* ---------- raytrace.comp:142-168 ---------- */
static int hitSphere(Ray ray, v3 center, float radius, float tMin, float tMax,
                     HitRecord *rec) {
  v3 oc = sub3(ray.origin, center);
  float half_b = dot3(oc, ray.direction);
  float c = fmaf(-radius, radius, dot3(oc, oc));
  float discriminant = fmaf(half_b, half_b, -c);
  if (discriminant < 0.0f) return 0;
  if (g_roots) g_roots[0]++;
  float sqrtDisc = sqrtf(discriminant);
  float t = -half_b - sqrtDisc;
  if (t < tMin || t > tMax) {
    t = -half_b + sqrtDisc;
    if (t < tMin || t > tMax) return 0;
  }
  rec->t = t;
  rec->point = madd3(ray.origin, t, ray.direction);
  rec->normal = scale3(sub3(rec->point, center), 1.0f / radius);
  return 1;
}

/* ---------- raytrace.comp:171-210 ---------- */
static int hitEllipsoid(Ray ray, v3 center, v3 radii, v4 rotation, float tMin,
                        float tMax, HitRecord *rec) {
  v3 oc_world = sub3(ray.origin, center);
  v3 oc = inverseRotateByQuaternion(oc_world, rotation);
  v3 dir = inverseRotateByQuaternion(ray.direction, rotation);
  v3 invRadii = V3(1.0f / radii.x, 1.0f / radii.y, 1.0f / radii.z);
  v3 ocScaled = mul3(oc, invRadii);
  v3 dirScaled = mul3(dir, invRadii);
  float a = dot3(dirScaled, dirScaled);
  float half_b = dot3(ocScaled, dirScaled);
  float c = dot3(ocScaled, ocScaled) - 1.0f;
  float discriminant = fmaf(half_b, half_b, -(a * c));
  if (discriminant < 0.0f) return 0;
  if (g_roots) g_roots[1]++;
  float sqrtDisc = sqrtf(discriminant);
  float t = (-half_b - sqrtDisc) / a;
  if (t < tMin || t > tMax) {
    t = (-half_b + sqrtDisc) / a;
    if (t < tMin || t > tMax) return 0;
  }
  rec->t = t;
  rec->point = madd3(ray.origin, t, ray.direction);
  v3 localPoint = madd3(oc, t, dir);
  v3 normalLocal = mul3(mul3(localPoint, invRadii), invRadii);
  rec->normal = normalize3(rotateByQuaternion(normalLocal, rotation));
  return 1;
}

/* ---------- raytrace.comp:212-241 (closest hit; ties go to the later object) ---------- */
static int traceRay(Ctx *c, Ray ray, HitRecord *rec, float tMin, float tMax) {
  float closestT = tMax;
  int hitAnything = 0;
  const PtPushConstants *pc = c->pc;
  c->cnt[PT_CNT_RAYS]++;
  c->cnt[PT_CNT_SPHERE_TESTS] += (uint64_t)(pc->numSpheres > 0 ? pc->numSpheres : 0);
  c->cnt[PT_CNT_ELLIPSOID_TESTS] += (uint64_t)(pc->numEllipsoids > 0 ? pc->numEllipsoids : 0);
  for (int i = 0; i < pc->numSpheres; i++) {
    /* updateScene repacks GPUSphere -> {center, radius, materialIndex}
     * (reference src/VulkanRenderer.cpp:803-822) */
    const PtGPUSphere *s = &c->s->spheres[i];
    HitRecord tmp;
    if (hitSphere(ray, ld3(s->center), s->radius, tMin, closestT, &tmp)) {
      closestT = tmp.t;
      *rec = tmp;
      rec->materialIndex = s->materialIndex;
      hitAnything = 1;
    }
  }
  for (int i = 0; i < pc->numEllipsoids; i++) {
    const PtGPUEllipsoid *e = &c->s->ellipsoids[i];
    v4 rot = {e->rotation[0], e->rotation[1], e->rotation[2], e->rotation[3]};
    HitRecord tmp;
    if (hitEllipsoid(ray, ld3(e->center), ld3(e->radii), rot, tMin, closestT, &tmp)) {
      closestT = tmp.t;
      *rec = tmp;
      rec->materialIndex = e->materialIndex;
      hitAnything = 1;
    }
  }
  return hitAnything;
}

/* ---------- raytrace.comp:243-253 ---------- */
static float getShadowFactor(Ctx *c, v3 point, v3 lightPos) {
  Ray shadowRay;
  shadowRay.origin = point;
  v3 lightVec = sub3(lightPos, point);
  float distToLight = length3(lightVec);
  shadowRay.direction = div3s(lightVec, distToLight);
  HitRecord rec;
  c->cnt[PT_CNT_LIGHT_RAYS]++;
  return traceRay(c, shadowRay, &rec, EPSILON, distToLight) ? 0.0f : 1.0f;
}

/* ---------- raytrace.comp:256-279 ---------- */
typedef struct {
  v3 hitPoint, normal, viewDir, surfaceColor;
  float ambient, diffuse, specular, shininess, emissiveStrength, time;
  int materialIndex;
} SurfaceShaderData;
typedef struct {
  v3 color, emissive;
  float emissiveStrength, ambient, diffuse, specular, shininess;
} SurfaceShaderResult;

/* ---------- rainbow.glsl:1-22 ---------- */
static float hue2rgb(float p, float q, float t) {
  if (t < 0.0f) t += 1.0f;
  if (t > 1.0f) t -= 1.0f;
  if (t < 1.0f / 6.0f) return p + ((q - p) * 6.0f) * t;
  if (t < 0.5f) return q;
  if (t < 2.0f / 3.0f) return p + ((q - p) * (2.0f / 3.0f - t)) * 6.0f;
  return p;
}
static v3 hsl2rgb(v3 hsl) {
  float H = gl_fract(hsl.x);
  float S = gl_clamp(hsl.y, 0.0f, 1.0f);
  float L = gl_clamp(hsl.z, 0.0f, 1.0f);
  if (S == 0.0f) return V3(L, L, L);
  float q = (L < 0.5f) ? (L * (1.0f + S)) : ((L + S) - L * S);
  float p = 2.0f * L - q;
  return V3(hue2rgb(p, q, H + 1.0f / 3.0f), hue2rgb(p, q, H),
            hue2rgb(p, q, H - 1.0f / 3.0f));
}
/* ---------- rainbow.glsl:24-50 (shader index 1) ---------- */
static SurfaceShaderResult surfaceShader_Rainbow(SurfaceShaderData data) {
  SurfaceShaderResult r;
  r.color = V3(0, 0, 0);
  r.emissive = V3(0, 0, 0);
  r.emissiveStrength = data.emissiveStrength;
  r.ambient = 0.0f;
  r.diffuse = 0.0f;
  r.specular = 0.0f;
  r.shininess = data.shininess;
  v3 sweepDir = normalize3(V3(1.0f, 0.5f, 0.25f));
  float freq = 2.0f, speed = 0.002f;
  float phase = dot3(data.hitPoint, sweepDir) * freq + data.time * speed;
  float cosVal = cos_f(phase);
  float hue = gl_fract(0.5f + 0.5f * cosVal);
  v3 emissiveColor = hsl2rgb(V3(hue, 1.0f, 0.5f));
  r.emissive = scale3(emissiveColor, r.emissiveStrength);
  return r;
}
/* ---------- toon.glsl:1-50 (shader index 2) ---------- */
static SurfaceShaderResult surfaceShader_Toon(Ctx *c, SurfaceShaderData data) {
  SurfaceShaderResult r;
  r.color = V3(0, 0, 0);
  r.emissive = V3(0, 0, 0);
  r.emissiveStrength = data.emissiveStrength;
  r.ambient = data.ambient;
  r.diffuse = data.diffuse;
  r.specular = data.specular;
  r.shininess = data.shininess;
  v3 finalColor = scale3(data.surfaceColor, data.ambient);
  for (int i = 0; i < c->pc->numLights; i++) {
    const PtGPULight *L = &c->s->lights[i];
    v3 lightPos = ld3(L->position);
    float intensity = L->intensity;
    float shadowFactor = getShadowFactor(c, data.hitPoint, lightPos);
    if (shadowFactor < 0.5f) continue;
    v3 lightVec = sub3(lightPos, data.hitPoint);
    float distance = length3(lightVec);
    v3 lightDir = div3s(lightVec, distance);
    float attenuation =
        intensity / ((1.0f + 0.09f * distance) + (0.032f * distance) * distance);
    float diffuseRaw = gl_max(0.0f, dot3(data.normal, lightDir));
    float diffuseQuantized = diffuseRaw > 0.6f ? 1.0f : (diffuseRaw > 0.3f ? 0.5f : 0.0f);
    v3 diffuse = scale3(scale3(data.surfaceColor, data.diffuse * diffuseQuantized), attenuation);
    v3 halfDir = normalize3(add3(lightDir, data.viewDir));
    float specularRaw = pow_f(gl_max(0.0f, dot3(data.normal, halfDir)), data.shininess);
    float specularQuantized = specularRaw > 0.8f ? 1.0f : 0.0f;
    v3 specular = scale3(scale3(scale3(data.surfaceColor, data.specular), specularQuantized), attenuation);
    finalColor = add3(finalColor, add3(diffuse, specular));
  }
  r.color = finalColor;
  if (data.emissiveStrength > 0.0f)
    r.emissive = scale3(data.surfaceColor, data.emissiveStrength);
  return r;
}

/* Generated dispatch chain (reference src/ShaderCompiler.cpp:131-185): files
 * sorted by path get indices 1, 2, ...: rainbow.glsl -> 1, toon.glsl -> 2.
 * An unknown index leaves the pre-initialised result untouched. */
static void dispatchSurfaceShader(Ctx *c, int index, SurfaceShaderData d,
                                  SurfaceShaderResult *r) {
  if (index == 1) *r = surfaceShader_Rainbow(d);
  else if (index == 2) *r = surfaceShader_Toon(c, d);
}

static SurfaceShaderData shaderDataFromMaterial(const PtGPUMaterial *m) {
  SurfaceShaderData d;
  memset(&d, 0, sizeof d);
  d.surfaceColor = ld3(m->colorAndAmbient);
  d.ambient = m->colorAndAmbient[3];
  d.diffuse = m->diffuseSpecularShiny[0];
  d.specular = m->diffuseSpecularShiny[1];
  d.shininess = m->diffuseSpecularShiny[2];
  d.emissiveStrength = m->transparencyEmissive[1];
  return d;
}
static SurfaceShaderResult shaderResultInit(const SurfaceShaderData *d) {
  SurfaceShaderResult r;
  r.color = V3(0, 0, 0);
  r.emissive = V3(0, 0, 0);
  r.emissiveStrength = d->emissiveStrength;
  r.ambient = d->ambient;
  r.diffuse = d->diffuse;
  r.specular = d->specular;
  r.shininess = d->shininess;
  return r;
}

/* ---------- raytrace.comp:286-483 ---------- */
static v3 calculateLighting(Ctx *c, const HitRecord *rec, v3 viewDir) {
  const PtPushConstants *pc = c->pc;
  const PtGPUMaterial *mat = &c->s->materials[rec->materialIndex];
  v3 surfaceColor = ld3(mat->colorAndAmbient);
  float ambient = mat->colorAndAmbient[3];
  float diffuseCoeff = mat->diffuseSpecularShiny[0];
  float specularCoeff = mat->diffuseSpecularShiny[1];
  float shininess = mat->diffuseSpecularShiny[2];
  float emissiveStrength = mat->transparencyEmissive[1];
  int shaderFunctionIndex = f2i(mat->transparencyEmissive[2]);
  c->cnt[PT_CNT_SHADED_HITS]++;

  if (shaderFunctionIndex != 0) { /* :297-326 */
    SurfaceShaderData d = shaderDataFromMaterial(mat);
    d.hitPoint = rec->point;
    d.normal = rec->normal;
    d.viewDir = viewDir;
    d.time = pc->time;
    d.materialIndex = rec->materialIndex;
    SurfaceShaderResult r = shaderResultInit(&d);
    dispatchSurfaceShader(c, shaderFunctionIndex, d, &r);
    return add3(r.color, r.emissive);
  }

  v3 finalColor = scale3(surfaceColor, ambient); /* :331 */
  if (emissiveStrength > 0.0f)
    finalColor = add3(finalColor, scale3(surfaceColor, emissiveStrength));

  for (int i = 0; i < pc->numLights; i++) { /* :339-364 */
    const PtGPULight *L = &c->s->lights[i];
    v3 lightPos = ld3(L->position); /* light buffer = vec4(position, intensity) */
    float intensity = L->intensity;
    float shadowFactor = getShadowFactor(c, rec->point, lightPos);
    if (shadowFactor < 0.5f) continue;
    v3 lightVec = sub3(lightPos, rec->point);
    float distance = length3(lightVec);
    v3 lightDir = div3s(lightVec, distance);
    float attenuation =
        intensity / ((1.0f + 0.09f * distance) + (0.032f * distance) * distance);
    float diffuseStrength = gl_max(0.0f, dot3(rec->normal, lightDir));
    /* diffuseCoeff * diffuseStrength * surfaceColor * attenuation */
    v3 diffuse = scale3(scale3(surfaceColor, diffuseCoeff * diffuseStrength), attenuation);
    v3 halfDir = normalize3(add3(lightDir, viewDir));
    float specularStrength = pow_f(gl_max(0.0f, dot3(rec->normal, halfDir)), shininess);
    /* surfaceColor * specularCoeff * specularStrength * attenuation */
    v3 specular = scale3(scale3(scale3(surfaceColor, specularCoeff), specularStrength), attenuation);
    finalColor = add3(finalColor, add3(diffuse, specular));
  }

  for (int i = 0; i < pc->numSpheres; i++) { /* :367-443 */
    const PtGPUSphere *sp = &c->s->spheres[i];
    const PtGPUMaterial *sm = &c->s->materials[sp->materialIndex];
    float sphereEmissive = sm->transparencyEmissive[1];
    int sphereShaderIndex = f2i(sm->transparencyEmissive[2]);
    if (sphereEmissive > 0.0f) {
      v3 center = ld3(sp->center);
      v3 emissiveColor = ld3(sm->colorAndAmbient);
      if (sphereShaderIndex != 0) {
        v3 toS = sub3(center, rec->point);
        float distS = length3(toS);
        v3 dirS = div3s(toS, distS);
        /* sphere.center - dirToSphere * sphere.radius */
        v3 approximateHitPoint = sub3(center, scale3(dirS, sp->radius));
        SurfaceShaderData d = shaderDataFromMaterial(sm);
        d.hitPoint = approximateHitPoint;
        d.normal = neg3(dirS);
        d.viewDir = dirS;
        d.emissiveStrength = sphereEmissive;
        d.time = pc->time;
        d.materialIndex = sp->materialIndex;
        SurfaceShaderResult r = shaderResultInit(&d);
        dispatchSurfaceShader(c, sphereShaderIndex, d, &r);
        emissiveColor = r.emissive;
        sphereEmissive = r.emissiveStrength;
        if (length3(emissiveColor) < 0.001f) continue;
      }
      v3 toSphere = sub3(center, rec->point);
      float distToSphere = length3(toSphere);
      v3 dirToSphere = div3s(toSphere, distToSphere);
      float alignment = dot3(rec->normal, dirToSphere);
      if (alignment > 0.0f) {
        Ray shadowRay;
        shadowRay.origin = madd3(rec->point, EPSILON, rec->normal);
        shadowRay.direction = dirToSphere;
        HitRecord shadowHit;
        float checkDist = (distToSphere - sp->radius) - EPSILON;
        c->cnt[PT_CNT_EMISSIVE_RAYS]++;
        int blocked = traceRay(c, shadowRay, &shadowHit, EPSILON, checkDist);
        if (!blocked) {
          float attenuation = sphereEmissive /
              ((1.0f + 0.5f * distToSphere) + (0.1f * distToSphere) * distToSphere);
          /* diffuseStrength * emissiveColor * attenuation */
          v3 emissiveLight = scale3(scale3(emissiveColor, alignment), attenuation);
          finalColor = add3(finalColor, emissiveLight);
        }
      }
    }
  }

  for (int i = 0; i < pc->numEllipsoids; i++) { /* :446-480 */
    const PtGPUEllipsoid *el = &c->s->ellipsoids[i];
    const PtGPUMaterial *em = &c->s->materials[el->materialIndex];
    float ellipsoidEmissive = em->transparencyEmissive[1];
    if (ellipsoidEmissive > 0.0f) {
      v3 emissiveColor = ld3(em->colorAndAmbient);
      v3 toE = sub3(ld3(el->center), rec->point);
      float distE = length3(toE);
      v3 dirE = div3s(toE, distE);
      float alignment = dot3(rec->normal, dirE);
      if (alignment > 0.0f) {
        Ray shadowRay;
        shadowRay.origin = madd3(rec->point, EPSILON, rec->normal);
        shadowRay.direction = dirE;
        HitRecord shadowHit;
        float approxRadius = gl_max(gl_max(el->radii[0], el->radii[1]), el->radii[2]);
        float checkDist = (distE - approxRadius) - EPSILON;
        c->cnt[PT_CNT_EMISSIVE_RAYS]++;
        int blocked = traceRay(c, shadowRay, &shadowHit, EPSILON, checkDist);
        if (!blocked) {
          float attenuation = ellipsoidEmissive /
              ((1.0f + 0.5f * distE) + (0.1f * distE) * distE);
          v3 emissiveLight = scale3(scale3(emissiveColor, alignment), attenuation);
          finalColor = add3(finalColor, emissiveLight);
        }
      }
    }
  }
  return finalColor;
}

/* ---------- raytrace.comp:485-517 (nearest neighbour) ---------- */
static float sampleVolume(Ctx *c, v3 worldPos, int volumeIndex) {
  if (volumeIndex >= c->pc->numVolumes) return 0.0f;
  const PtGPUVolumetricData *vol = &c->s->volumes[volumeIndex];
  v3 volPos = ld3(vol->position);
  v3 v0 = ld3(vol->v0), v1 = ld3(vol->v1);
  int res_x = vol->resolution_x, res_y = vol->resolution_y, res_z = vol->resolution_z;
  v3 relPos = sub3(worldPos, volPos);
  if (relPos.x < v0.x || relPos.y < v0.y || relPos.z < v0.z ||
      relPos.x > v1.x || relPos.y > v1.y || relPos.z > v1.z)
    return 0.0f;
  v3 ext = sub3(v1, v0);
  v3 d = sub3(relPos, v0);
  relPos = V3(d.x / ext.x, d.y / ext.y, d.z / ext.z);
  int ix = f2i(relPos.x * (float)res_x);
  int iy = f2i(relPos.y * (float)res_y);
  int iz = f2i(relPos.z * (float)res_z);
  ix = ix < 0 ? 0 : (ix > res_x - 1 ? res_x - 1 : ix);
  iy = iy < 0 ? 0 : (iy > res_y - 1 ? res_y - 1 : iy);
  iz = iz < 0 ? 0 : (iz > res_z - 1 ? res_z - 1 : iz);
  int index = ix + iy * res_x + iz * res_x * res_y;
  /* byte (index % 4) of little-endian uint voxelData[index / 4] == byte[index];
   * bytes past the uploaded data read as 0 */
  uint32_t byteValue = 0;
  if (index >= 0 && (uint64_t)index < c->s->voxel_bytes && c->s->voxels)
    byteValue = c->s->voxels[index];
  if (c->volume_filter == 1) {
    /* Trilinear extension (no counterpart in the reference): voxel centres sit at
     * (i + 0.5) / res; the sample point in voxel units is u * res - 0.5, clamped to
     * [0, res - 1]; lerp(a, b, t) = fma(t, b - a, a) along x, then y, then z, on the
     * byte values as floats; the result is divided by 255 like the nearest sample. */
    float p[3] = {relPos.x * (float)res_x - 0.5f, relPos.y * (float)res_y - 0.5f, relPos.z * (float)res_z - 0.5f};
    int res[3] = {res_x, res_y, res_z};
    int i0[3], i1[3];
    float f[3];
    for (int a = 0; a < 3; a++) {
      float hi = (float)(res[a] - 1);
      float q = p[a] < 0.0f ? 0.0f : (p[a] > hi ? hi : p[a]);
      if (q != q) q = 0.0f;
      i0[a] = (int)q;
      f[a] = q - (float)i0[a];
      i1[a] = i0[a] + 1 > res[a] - 1 ? res[a] - 1 : i0[a] + 1;
    }
    float b[2][2][2];
    for (int z = 0; z < 2; z++)
      for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
          int64_t id = (int64_t)(x ? i1[0] : i0[0]) + (int64_t)(y ? i1[1] : i0[1]) * res_x +
                       (int64_t)(z ? i1[2] : i0[2]) * res_x * res_y;
          uint32_t v = 0;
          if (id >= 0 && (uint64_t)id < c->s->voxel_bytes && c->s->voxels) v = c->s->voxels[id];
          b[z][y][x] = (float)v;
        }
    float c00 = fmaf(f[0], b[0][0][1] - b[0][0][0], b[0][0][0]);
    float c10 = fmaf(f[0], b[0][1][1] - b[0][1][0], b[0][1][0]);
    float c01 = fmaf(f[0], b[1][0][1] - b[1][0][0], b[1][0][0]);
    float c11 = fmaf(f[0], b[1][1][1] - b[1][1][0], b[1][1][0]);
    float c0 = fmaf(f[1], c10 - c00, c00);
    float c1 = fmaf(f[1], c11 - c01, c01);
    return fmaf(f[2], c1 - c0, c0) / 255.0f;
  }
  return (float)byteValue / 255.0f;
}

/* ---------- raytrace.comp:519-532 ---------- */
static int intersectBox(Ray ray, v3 boxMin, v3 boxMax, float *tNear, float *tFar) {
  v3 invDir = V3(1.0f / (ray.direction.x + 1e-8f), 1.0f / (ray.direction.y + 1e-8f),
                 1.0f / (ray.direction.z + 1e-8f));
  v3 t0 = mul3(sub3(boxMin, ray.origin), invDir);
  v3 t1 = mul3(sub3(boxMax, ray.origin), invDir);
  v3 tmin = V3(gl_min(t0.x, t1.x), gl_min(t0.y, t1.y), gl_min(t0.z, t1.z));
  v3 tmax = V3(gl_max(t0.x, t1.x), gl_max(t0.y, t1.y), gl_max(t0.z, t1.z));
  *tNear = gl_max(gl_max(tmin.x, tmin.y), tmin.z);
  *tFar = gl_min(gl_min(tmax.x, tmax.y), tmax.z);
  return *tFar >= *tNear && *tFar > 0.0f;
}

/* ---------- raytrace.comp:534-585 ---------- */
static v3 rayMarchVolume(Ctx *c, Ray ray, float maxDist, int volumeIndex) {
  if (c->pc->numVolumes <= 0) return V3(-1.0f, -1.0f, -1.0f);
  const PtGPUVolumetricData *vol = &c->s->volumes[volumeIndex];
  v3 volPos = ld3(vol->position);
  v3 v0 = ld3(vol->v0), v1 = ld3(vol->v1);
  int materialIndex = vol->materialIndex;
  v3 boxMin = add3(volPos, v0);
  v3 boxMax = add3(volPos, v1);
  float tNear, tFar;
  if (!intersectBox(ray, boxMin, boxMax, &tNear, &tFar)) return V3(-1.0f, -1.0f, -1.0f);
  tNear = gl_max(tNear, 0.0f);
  tFar = gl_min(tFar, maxDist);
  const PtGPUMaterial *mat = &c->s->materials[materialIndex];
  v3 materialColor = ld3(mat->scatterAndAbsorption);
  float absorptionCoeff = mat->scatterAndAbsorption[3];
  float stepSize = 0.02f;
  float t = tNear;
  float accumulatedDensity = 0.0f;
  int stepCount = 0;
  const int maxSteps = 500;
  const float TRANSPARENCY_EPSILON = 0.0001f;
  while (t < tFar && stepCount < maxSteps) {
    v3 samplePos = madd3(ray.origin, t, ray.direction);
    float density = sampleVolume(c, samplePos, volumeIndex);
    float sampleAlpha = density * absorptionCoeff;
    accumulatedDensity = fmaf(1.0f - accumulatedDensity, sampleAlpha, accumulatedDensity);
    t += stepSize;
    stepCount++;
  }
  c->cnt[PT_CNT_VOXEL_SAMPLES] += (uint64_t)stepCount;
  /* mix(vec3(-1), materialColor*acc, float(acc >= eps)) = x*(1-a) + y*a */
  float a = (accumulatedDensity >= TRANSPARENCY_EPSILON) ? 1.0f : 0.0f;
  v3 y = scale3(materialColor, accumulatedDensity);
  float oma = 1.0f - a;
  return V3(-1.0f * oma + y.x * a, -1.0f * oma + y.y * a, -1.0f * oma + y.z * a);
}

/* ---------- raytrace.comp:595-684 ---------- */
static v3 traceIterative(Ctx *c, Ray initialRay) {
  const PtPushConstants *pc = c->pc;
  v3 finalColor = V3(0, 0, 0);
  v3 throughput = V3(1.0f, 1.0f, 1.0f);
  const float THROUGHPUT_THRESHOLD = 0.001f;
  Ray currentRay = initialRay;
  float tMin = EPSILON;
  for (int bounce = 0; bounce < c->max_bounces; bounce++) {
    if (length3(throughput) < THROUGHPUT_THRESHOLD) break;
    HitRecord rec;
    c->cnt[PT_CNT_PATH_RAYS]++;
    int hitSolid = traceRay(c, currentRay, &rec, tMin, MAX_DISTANCE);
    v3 volumetricContribution = V3(-1.0f, -1.0f, -1.0f);
    float volumeAlpha = 0.0f;
    if (pc->numVolumes > 0) {
      volumetricContribution =
          rayMarchVolume(c, currentRay, hitSolid ? rec.t : MAX_DISTANCE, 0);
      if (volumetricContribution.x >= 0.0f) {
        volumeAlpha = gl_clamp(length3(volumetricContribution) /
                                   length3(V3(1.0f, 1.0f, 1.0f)), 0.0f, 1.0f);
      }
    }
    if (volumeAlpha > 0.0f) {
      finalColor = add3(finalColor, scale3(mul3(throughput, volumetricContribution), volumeAlpha));
      throughput = scale3(throughput, 1.0f - volumeAlpha);
    }
    if (hitSolid) {
      const PtGPUMaterial *mat = &c->s->materials[rec.materialIndex];
      float emissiveStrength = mat->transparencyEmissive[1];
      v3 surfaceColor = ld3(mat->colorAndAmbient);
      if (emissiveStrength > 0.0f)
        finalColor = add3(finalColor, scale3(mul3(throughput, surfaceColor), emissiveStrength));
      v3 viewDir = normalize3(sub3(currentRay.origin, rec.point));
      v3 litColor = calculateLighting(c, &rec, viewDir);
      float transparency = mat->transparencyEmissive[0];
      float reflectivity = mat->diffuseSpecularShiny[3];
      if (reflectivity > 0.01f) {
        float surfaceOpacity = (1.0f - reflectivity) * (1.0f - transparency);
        finalColor = add3(finalColor, scale3(mul3(throughput, litColor), surfaceOpacity));
        throughput = scale3(throughput, reflectivity);
        /* reflect(I, N) = I - 2*dot(N, I)*N */
        float k = 2.0f * dot3(rec.normal, currentRay.direction);
        v3 reflectDir = sub3(currentRay.direction, scale3(rec.normal, k));
        currentRay.origin = madd3(rec.point, EPSILON, rec.normal);
        currentRay.direction = reflectDir;
        tMin = EPSILON;
      } else if (transparency > 0.01f) {
        float opacity = 1.0f - transparency;
        finalColor = add3(finalColor, scale3(mul3(throughput, litColor), opacity));
        throughput = scale3(throughput, transparency);
        currentRay.origin = madd3(rec.point, EPSILON, currentRay.direction);
        tMin = EPSILON;
      } else {
        finalColor = add3(finalColor, mul3(throughput, litColor));
        break;
      }
    } else {
      v3 unitDir = normalize3(currentRay.direction);
      float t = 0.5f * (unitDir.y + 1.0f);
      float omt = 1.0f - t;
      v3 bot = ld3(pc->bgColorBottom), top = ld3(pc->bgColorTop);
      v3 bgColor = V3(bot.x * omt + top.x * t, bot.y * omt + top.y * t, bot.z * omt + top.z * t);
      finalColor = add3(finalColor, mul3(throughput, bgColor));
      break;
    }
  }
  return finalColor;
}

static inline uint8_t unorm8(float c) {
  if (c != c) return 0;
  return (uint8_t)(int)rintf(c * 255.0f);
}

/* Sub-pixel offset of sample `s` of pixel `pixel` (= y * width + x), in [0, 1)^2.
 * Sample 0 is the reference's pixel-corner sample; the others come from a
 * counter-based integer hash (extension, no counterpart in the reference). */
void pto_sample_offset(uint32_t pixel, uint32_t s, float *jx, float *jy) {
  if (s == 0) { *jx = 0.0f; *jy = 0.0f; return; }
  uint32_t h = pixel * 0x9E3779B1u + s * 0x85EBCA77u;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  *jx = (float)(h & 0xFFFFu) / 65536.0f;
  *jy = (float)(h >> 16) / 65536.0f;
}

/* ---------- raytrace.comp:686-705 ---------- */
static void renderPixel(Ctx *c, int px, int py, float *out_f32, uint8_t *out_rgba8) {
  v3 color = V3(0, 0, 0);
  int n = c->num_samples > 1 ? c->num_samples : 1;
  v3 sum = V3(0, 0, 0);
  for (int k = 0; k < n; k++) {
    float jx, jy;
    pto_sample_offset((uint32_t)py * (uint32_t)c->width + (uint32_t)px, (uint32_t)(c->first_sample + k), &jx, &jy);
    float uvx = ((float)px + jx) / (float)c->width;
    float uvy = ((float)py + jy) / (float)c->height;
    uvy = 1.0f - uvy;
    Ray ray = createCameraRay(c, uvx, uvy);
    color = traceIterative(c, ray);
    color = V3(gl_clamp(color.x, 0.0f, 1.0f), gl_clamp(color.y, 0.0f, 1.0f),
               gl_clamp(color.z, 0.0f, 1.0f));
    sum = add3(sum, color);
  }
  /* accumulation: running sum of the clamped samples, divided by the count */
  if (n > 1) color = V3(sum.x / (float)n, sum.y / (float)n, sum.z / (float)n);
  size_t i = ((size_t)py * (size_t)c->width + (size_t)px) * 4;
  if (out_f32) {
    out_f32[i + 0] = color.x; out_f32[i + 1] = color.y;
    out_f32[i + 2] = color.z; out_f32[i + 3] = 1.0f;
  }
  if (out_rgba8) {
    out_rgba8[i + 0] = unorm8(color.x); out_rgba8[i + 1] = unorm8(color.y);
    out_rgba8[i + 2] = unorm8(color.z); out_rgba8[i + 3] = 255;
  }
}

typedef struct {
  Ctx ctx;
  int tid, nthreads, row_begin, row_end;
  float *out_f32;
  uint8_t *out_rgba8;
} Job;

static void *worker(void *arg) {
  Job *j = (Job *)arg;
  g_roots = &j->ctx.cnt[PT_CNT_SPHERE_ROOTS];
  for (int y = j->row_begin + j->tid; y < j->row_end; y += j->nthreads)
    for (int x = 0; x < j->ctx.width; x++)
      renderPixel(&j->ctx, x, y, j->out_f32, j->out_rgba8);
  g_roots = 0;
  return NULL;
}

int pto_counter_count(void) { return PT_CNT_COUNT; }

int pto_render(const PtoScene *scene, const PtPushConstants *pc, int width,
               int height, int max_bounces, int row_begin, int row_end,
               float *out_f32, uint8_t *out_rgba8, uint64_t *counters,
               int nthreads) {
  return pto_render_ex(scene, pc, width, height, max_bounces, row_begin, row_end, out_f32, out_rgba8,
                       counters, nthreads, NULL);
}

int pto_render_ex(const PtoScene *scene, const PtPushConstants *pc, int width,
                  int height, int max_bounces, int row_begin, int row_end,
                  float *out_f32, uint8_t *out_rgba8, uint64_t *counters,
                  int nthreads, const PtoOptions *opt) {
  if (nthreads < 1) nthreads = 1;
  if (row_begin < 0) row_begin = 0;
  if (row_end > height) row_end = height;
  Job *jobs = (Job *)calloc((size_t)nthreads, sizeof(Job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; t++) {
    jobs[t].ctx.s = scene;
    jobs[t].ctx.pc = pc;
    jobs[t].ctx.width = width;
    jobs[t].ctx.height = height;
    jobs[t].ctx.max_bounces = max_bounces;
    jobs[t].ctx.volume_filter = opt ? opt->volume_filter : 0;
    jobs[t].ctx.first_sample = opt ? opt->first_sample : 0;
    jobs[t].ctx.num_samples = opt ? opt->num_samples : 1;
    jobs[t].tid = t;
    jobs[t].nthreads = nthreads;
    jobs[t].row_begin = row_begin;
    jobs[t].row_end = row_end;
    jobs[t].out_f32 = out_f32;
    jobs[t].out_rgba8 = out_rgba8;
  }
  if (nthreads == 1) {
    worker(&jobs[0]);
  } else {
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  }
  if (counters) {
    for (int k = 0; k < PT_CNT_COUNT; k++) {
      counters[k] = 0;
      for (int t = 0; t < nthreads; t++) counters[k] += jobs[t].ctx.cnt[k];
    }
  }
  free(jobs);
  free(th);
  return 0;
}

/* ---------- probes ---------- */
static void st3(float *p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

int pto_hit_sphere(const float origin[3], const float dir[3], const float center[3],
                   float radius, float t_min, float t_max, float *t, float *point,
                   float *normal) {
  Ray r = {ld3(origin), ld3(dir)};
  HitRecord rec;
  if (!hitSphere(r, ld3(center), radius, t_min, t_max, &rec)) return 0;
  *t = rec.t; st3(point, rec.point); st3(normal, rec.normal);
  return 1;
}

int pto_hit_ellipsoid(const float origin[3], const float dir[3], const float center[3],
                      const float radii[3], const float rotation[4], float t_min,
                      float t_max, float *t, float *point, float *normal) {
  Ray r = {ld3(origin), ld3(dir)};
  v4 q = {rotation[0], rotation[1], rotation[2], rotation[3]};
  HitRecord rec;
  if (!hitEllipsoid(r, ld3(center), ld3(radii), q, t_min, t_max, &rec)) return 0;
  *t = rec.t; st3(point, rec.point); st3(normal, rec.normal);
  return 1;
}

void pto_rotate(const float v[3], const float q[4], int inverse, float out[3]) {
  v4 qq = {q[0], q[1], q[2], q[3]};
  st3(out, inverse ? inverseRotateByQuaternion(ld3(v), qq) : rotateByQuaternion(ld3(v), qq));
}

void pto_camera_ray(const float camera_pos[3], int width, int height, int px, int py,
                    float origin[3], float dir[3]) {
  PtPushConstants pc;
  memset(&pc, 0, sizeof pc);
  memcpy(pc.cameraPos, camera_pos, 12);
  Ctx c;
  memset(&c, 0, sizeof c);
  c.pc = &pc; c.width = width; c.height = height;
  float uvx = (float)px / (float)width;
  float uvy = 1.0f - (float)py / (float)height;
  Ray r = createCameraRay(&c, uvx, uvy);
  st3(origin, r.origin); st3(dir, r.direction);
}

void pto_march_volume(const PtoScene *scene, const PtPushConstants *pc,
                      const float origin[3], const float dir[3], float max_dist,
                      float out[3], uint64_t *steps) {
  Ctx c;
  memset(&c, 0, sizeof c);
  c.s = scene; c.pc = pc;
  Ray r = {ld3(origin), ld3(dir)};
  st3(out, rayMarchVolume(&c, r, max_dist, 0));
  if (steps) *steps = c.cnt[PT_CNT_VOXEL_SAMPLES];
}

int pto_intersect_box(const float origin[3], const float dir[3], const float box_min[3],
                      const float box_max[3], float *t_near, float *t_far) {
  Ray r = {ld3(origin), ld3(dir)};
  return intersectBox(r, ld3(box_min), ld3(box_max), t_near, t_far);
}

float pto_sample_volume(const PtoScene *scene, const PtPushConstants *pc, const float point[3],
                        int volume_filter) {
  Ctx c;
  memset(&c, 0, sizeof c);
  c.s = scene; c.pc = pc; c.volume_filter = volume_filter;
  return sampleVolume(&c, ld3(point), 0);
}

void pto_surface_shader(const PtoScene *scene, const PtPushConstants *pc, int shader_index,
                        const float hit_point[3], const float normal[3],
                        const float view_dir[3], int material_index, float out[3]) {
  Ctx c;
  memset(&c, 0, sizeof c);
  c.s = scene; c.pc = pc;
  const PtGPUMaterial *m = &scene->materials[material_index];
  SurfaceShaderData d = shaderDataFromMaterial(m);
  d.hitPoint = ld3(hit_point); d.normal = ld3(normal); d.viewDir = ld3(view_dir);
  d.time = pc->time; d.materialIndex = material_index;
  SurfaceShaderResult r = shaderResultInit(&d);
  dispatchSurfaceShader(&c, shader_index, d, &r);
  st3(out, add3(r.color, r.emissive));
}
