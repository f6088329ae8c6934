// host_probe.cpp — exercises the C++ scene API from pytest: reads one command
// per line on stdin and prints the result with 17 significant digits.
#include "path_tracer/Camera.h"
#include "path_tracer/Ellipsoid.h"
#include "path_tracer/HipRenderer.h"
#include "path_tracer/Quaternion.h"
#include "path_tracer/SceneWrappers.h"
#include "path_tracer/VolumeLoader.h"

#include <cstddef>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

static void pq(const Quaternion &q) { std::printf("%.17g %.17g %.17g %.17g\n", q.w, q.x, q.y, q.z); }
static void pv(const Vec3 &v) { std::printf("%.17g %.17g %.17g\n", v.x, v.y, v.z); }

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string cmd;
    in >> cmd;
    if (cmd == "axisangle") {
      double x, y, z, a; in >> x >> y >> z >> a;
      pq(Quaternion::fromAxisAngle(Vec3(x, y, z), a));
    } else if (cmd == "mul" || cmd == "slerp" || cmd == "lerp") {
      double a[4], b[4], t = 0; for (double &v : a) in >> v; for (double &v : b) in >> v; in >> t;
      Quaternion q1(a[0], a[1], a[2], a[3]), q2(b[0], b[1], b[2], b[3]);
      pq(cmd == "mul" ? q1 * q2 : (cmd == "slerp" ? Quaternion::slerp(q1, q2, t) : Quaternion::lerp(q1, q2, t)));
    } else if (cmd == "fromeuler") {
      double r, p, y; in >> r >> p >> y;
      pq(Quaternion::fromEulerAngles(r, p, y));
    } else if (cmd == "toeuler") {
      double a[4]; for (double &v : a) in >> v;
      double r, p, y; Quaternion(a[0], a[1], a[2], a[3]).toEulerAngles(r, p, y);
      std::printf("%.17g %.17g %.17g\n", r, p, y);
    } else if (cmd == "axis_angle_of") {
      double a[4]; for (double &v : a) in >> v;
      Quaternion q(a[0], a[1], a[2], a[3]); Vec3 ax = q.getAxis();
      std::printf("%.17g %.17g %.17g %.17g\n", ax.x, ax.y, ax.z, q.getAngle());
    } else if (cmd == "rotate") { // q given as x y z w
      double q[4], v[3]; for (double &x : q) in >> x; for (double &x : v) in >> x;
      Quaternion qq(q[3], q[0], q[1], q[2]);
      pv(qq.rotateVector(Vec3(v[0], v[1], v[2])));
    } else if (cmd == "rotmat_roundtrip") { // axis-angle -> matrix (via rotateVector of the basis) -> quaternion
      double x, y, z, a; in >> x >> y >> z >> a;
      Quaternion q = Quaternion::fromAxisAngle(Vec3(x, y, z), a);
      Vec3 c0 = q.rotateVector(Vec3(1, 0, 0)), c1 = q.rotateVector(Vec3(0, 1, 0)), c2 = q.rotateVector(Vec3(0, 0, 1));
      Quaternion r = Quaternion::fromRotationMatrix(c0.x, c1.x, c2.x, c0.y, c1.y, c2.y, c0.z, c1.z, c2.z);
      if (Quaternion::dot(q, r) < 0) r = r * -1.0;
      pq(r); pq(q);
    } else if (cmd == "camera_ray") {
      double e[3], asp, u, v; for (double &x : e) in >> x; in >> asp >> u >> v;
      Camera cam(Vec3(e[0], e[1], e[2]), Vec3(2, 1.5, 0), Vec3(0, 1, 0), 60.0, asp);
      pv(cam.getRay(u, v).direction);
    } else if (cmd == "layout") {
      std::printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(GPUSphere), sizeof(GPUEllipsoid), sizeof(GPUMaterial),
                  sizeof(GPULight), sizeof(GPUVolumetricData), sizeof(PushConstants), offsetof(PushConstants, cameraPos),
                  offsetof(PushConstants, bgColorTop), offsetof(PushConstants, bgColorBottom));
    } else if (cmd == "shaders") {
      for (int i = 0; i < pt_shader_count(); ++i) std::printf("%s %d\n", pt_shader_path(i), pt_shader_index(pt_shader_path(i)));
      std::printf("unknown %d\n", pt_shader_index("shaders/surface_shaders/none.glsl"));
    } else if (cmd == "materials") {
      // the demo's pre-render pass: dedupe by address, first-seen over spheres -> ellipsoids -> volumes
      Material red = Material::Diffuse(glm::vec3(0.8f, 0.2f, 0.2f), 0.7f, 0.1f);
      Material green = Material::Emissive(glm::vec3(0.2f, 0.8f, 0.2f), 2.0f);
      Material blue = Material::CustomShader(glm::vec3(0.2f, 0.2f, 0.8f), "shaders/surface_shaders/rainbow.glsl");
      Material mirror = Material::Mirror(glm::vec3(0.9f, 0.9f, 0.9f), 0.9f);
      Material vol = Material::Volumetric(glm::vec3(0.8f, 0.6f, 0.4f), 8.0f);
      Material toon = Material::CustomShader(glm::vec3(1.0f), "shaders/surface_shaders/toon.glsl");
      Material bogus = Material::CustomShader(glm::vec3(1.0f), "nope.glsl");
      std::vector<Sphere> s = {Sphere(glm::vec3(0.0f), 1.0f, glm::vec3(1.0f), red), Sphere(glm::vec3(0.0f), 1.0f, glm::vec3(1.0f), green),
                               Sphere(glm::vec3(0.0f), 1.0f, glm::vec3(1.0f), red), Sphere()};
      std::vector<Ellipsoid> e = {Ellipsoid(glm::vec3(0.0f), glm::vec3(1.0f), glm::vec3(1.0f), mirror),
                                  Ellipsoid(glm::vec3(0.0f), glm::vec3(1.0f), glm::vec3(1.0f), blue)};
      std::vector<VolumetricData> v = {VolumetricData(glm::vec3(0.0f), 1.0f, glm::vec3(0.0f), glm::vec3(1.0f), glm::ivec3(2, 3, 4), vol)};
      std::vector<Material> mats;
      SceneManager::prepareForRender(mats, s, e, v);
      std::printf("%zu", mats.size());
      for (auto &x : s) std::printf(" %d", x.getMaterialIndex());
      for (auto &x : e) std::printf(" %d", x.getMaterialIndex());
      for (auto &x : v) std::printf(" %d", x.getMaterialIndex());
      std::printf("\n");
      std::unordered_map<std::string, int> map;
      for (int i = 0; i < pt_shader_count(); ++i) map[pt_shader_path(i)] = i + 1;
      for (const Material *m : {&red, &green, &blue, &mirror, &vol, &toon, &bogus}) {
        GPUMaterial g = m->toGPU(map);
        const float *f = reinterpret_cast<const float *>(&g);
        for (int i = 0; i < 20; ++i) std::printf("%.9g ", f[i]);
        std::printf("\n");
      }
      GPUVolumetricData gv = v[0].toGPU();
      std::printf("%d %d %d %d\n", gv.resolution_x, gv.resolution_y, gv.resolution_z, gv.materialIndex);
    } else if (cmd == "loadvolume") {
      std::string dat, raw; in >> dat >> raw;
      std::vector<uint8_t> data; glm::ivec3 res;
      bool ok = loadVolumetricData(dat, raw, data, res);
      unsigned long sum = 0; for (uint8_t b : data) sum += b;
      std::printf("%d %d %d %d %zu %lu\n", ok ? 1 : 0, res.x, res.y, res.z, data.size(), sum);
    } else if (!cmd.empty()) {
      std::printf("unknown command %s\n", cmd.c_str());
    }
  }
  return 0;
}
