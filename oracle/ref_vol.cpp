// ref_vol.cpp — golden vectors from the REFERENCE's own legacy volume class.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile (target `ref`) against the reference's
// sources where they lie (/root/reference/src/{Vec3,Ray,Object,Material,Volumetric}.cpp, none of
// which need Vulkan/GLM/SDL); the binary goes to oracle/_ref/ and is never committed.  Its JSON
// output is committed as tests/golden/ref_volumetric.json.
//
// What the reference functions pin:
//   Volumetric::Volumetric  (reference src/Volumetric.cpp:7-71)   the .dat header parser ("Resolution",
//                           "SliceThickness"; tabs, spaces, CR LF) and the bounds it derives; run on the
//                           reference's volume/walnut.dat and on variants of its text
//   Volumetric::hit         (reference src/Volumetric.cpp:73-153) a slab test of a ray against the box; for
//                           rays that start outside the box and are not parallel to a face its entry
//                           distance is the shader's tNear (shaders/raytrace.comp:519-532)
//   Volumetric::getDensity  (reference src/Volumetric.cpp:155-192) floor-indexed nearest voxel / 255: for a
//                           box that starts at the volume's position this is the voxel the shader's
//                           sampleVolume reads (shaders/raytrace.comp:485-517)
// The class prints to std::cout while it loads; the generator captures that text and reads the parsed
// resolution, thickness and bounds from it (the members are private).
#include "Ray.h"
#include "Vec3.h"
#include "Volumetric.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

static uint64_t g_state = 0x2545f4914f6cdd1dULL;
static uint32_t rnd32() { // splitmix64, high half
  uint64_t z = (g_state += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}
static float rndf(float lo, float hi) {
  float u = (float)(rnd32() >> 8) * (1.0f / 16777216.0f);
  return lo + (hi - lo) * u;
}

static std::string json_escape(const std::string &s) {
  std::string o;
  char buf[8];
  for (unsigned char c : s) {
    if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
    else if (c == '\n') o += "\\n";
    else if (c == '\r') o += "\\r";
    else if (c == '\t') o += "\\t";
    else if (c < 0x20) { std::snprintf(buf, sizeof buf, "\\u%04x", c); o += buf; }
    else o += (char)c;
  }
  return o;
}

struct Loaded {
  std::unique_ptr<Volumetric> vol;
  std::string log;
};
static Loaded load(const std::string &dat, const std::string &raw, Vec3 pos, double scale) {
  std::ostringstream out, err;
  std::streambuf *o = std::cout.rdbuf(out.rdbuf()), *e = std::cerr.rdbuf(err.rdbuf());
  Loaded l;
  l.vol.reset(new Volumetric(dat, raw, pos, scale, std::shared_ptr<Material>()));
  std::cout.rdbuf(o);
  std::cerr.rdbuf(e);
  l.log = out.str() + err.str();
  return l;
}
// "Key: a b c" as the class printed it (absent: the header line was not recognised)
static bool logged3(const std::string &log, const char *key, double v[3]) {
  size_t at = log.find(std::string("\n") + key);
  if (at == std::string::npos) return false;
  return std::sscanf(log.c_str() + at + 1 + std::string(key).size(), " %lf %lf %lf", &v[0], &v[1], &v[2]) == 3;
}

int main(int argc, char **argv) {
  const std::string refRoot = argc > 1 ? argv[1] : "/root/reference";
  char tmpl[] = "/tmp/ref_vol.XXXXXX";
  const char *dir = mkdtemp(tmpl);
  if (!dir) return 1;
  const std::string datPath = std::string(dir) + "/v.dat", rawPath = std::string(dir) + "/v.raw";

  std::printf("{\n");
  // ---- the .dat header parser ----
  std::vector<std::string> texts;
  {
    std::ifstream f(refRoot + "/volume/walnut.dat", std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    texts.push_back(ss.str()); // the reference's own header, byte for byte
  }
  texts.push_back("Resolution: 4 3 2\n");
  texts.push_back("ObjectFileName:\tv.raw\r\nResolution:\t7 5 3\r\nSliceThickness:\t2 1 0.5\r\nFormat:\t\tUCHAR\r\n");
  texts.push_back("ObjectFileName: x\nResolution:40 29 35\nSliceThickness: 1 1 1\n");
  texts.push_back("Resolution:   12\t 6   9   \n");
  texts.push_back("Format: UCHAR\n");                   // no resolution at all
  texts.push_back("  Resolution: 9 9 9\n");             // leading blanks: not the key "Resolution"
  texts.push_back("Resolution : 6 5 4\n");              // blank before the colon: the key is trimmed
  texts.push_back("resolution: 5 5 5\n");               // other case: not the key
  texts.push_back("Resolution: 8 8\n");                 // two numbers only
  std::printf("\"dat_headers\": [\n");
  for (size_t i = 0; i < texts.size(); ++i) {
    { std::ofstream f(datPath, std::ios::binary); f << texts[i]; }
    { std::ofstream f(rawPath, std::ios::binary); f << "abc"; }
    Loaded l = load(datPath, rawPath, Vec3(0, 0, 0), 1.0);
    double res[3] = {0, 0, 0}, th[3] = {0, 0, 0};
    const bool hasRes = logged3("\n" + l.log, "Resolution:", res), hasTh = logged3("\n" + l.log, "SliceThickness:", th);
    std::printf("{\"text\": \"%s\", \"resolution_line_recognised\": %d, \"resolution\": [%d, %d, %d], "
                "\"thickness_line_recognised\": %d, \"thickness\": [%.17g, %.17g, %.17g]}%s\n",
                json_escape(texts[i]).c_str(), hasRes ? 1 : 0, (int)res[0], (int)res[1], (int)res[2], hasTh ? 1 : 0, th[0], th[1],
                th[2], i + 1 < texts.size() ? "," : "");
  }
  std::printf("],\n");

  // ---- one small volume for the slab test and the voxel lookup ----
  const int R[3] = {20, 12, 16};
  const double scale = 0.1;
  const Vec3 pos(1.5, 1.0, -0.5);
  {
    std::ofstream f(datPath, std::ios::binary);
    f << "ObjectFileName:\tv.raw\r\nResolution:\t" << R[0] << " " << R[1] << " " << R[2] << "\r\nSliceThickness:\t1 1 1\r\nFormat:\t\tUCHAR\r\n";
  }
  std::vector<uint8_t> bytes((size_t)R[0] * R[1] * R[2]);
  for (size_t i = 0; i < bytes.size(); ++i) bytes[i] = (uint8_t)(((uint32_t)i * 2654435761u) >> 24);
  { std::ofstream f(rawPath, std::ios::binary); f.write((const char *)bytes.data(), (std::streamsize)bytes.size()); }
  Loaded small = load(datPath, rawPath, pos, scale);
  const Vec3 lo = pos, hi(pos.x + R[0] * scale, pos.y + R[1] * scale, pos.z + R[2] * scale);
  std::printf("\"volume\": {\"resolution\": [%d, %d, %d], \"scale\": %.17g, \"position\": [%.17g, %.17g, %.17g], "
              "\"box_max\": [%.17g, %.17g, %.17g], \"voxel_rule\": \"byte i = ((i * 2654435761) mod 2^32) >> 24, x fastest\"},\n",
              R[0], R[1], R[2], scale, pos.x, pos.y, pos.z, hi.x, hi.y, hi.z);

  // ---- Volumetric::hit ----
  std::printf("\"slab_hit\": [\n");
  const int NH = 240;
  for (int i = 0; i < NH; ++i) {
    const char *kind = "outside";
    Vec3 o(rndf(-4, 8), rndf(-3, 5), rndf(-5, 7));
    // aim at a point of a box twice the size, so that about half of the rays miss
    Vec3 aim(lo.x + (hi.x - lo.x) * rndf(-0.5f, 1.5f), lo.y + (hi.y - lo.y) * rndf(-0.5f, 1.5f), lo.z + (hi.z - lo.z) * rndf(-0.5f, 1.5f));
    if (i % 12 == 10) { // origin inside the box
      o = Vec3((float)(lo.x + (hi.x - lo.x) * rndf(0.1f, 0.9f)), (float)(lo.y + (hi.y - lo.y) * rndf(0.1f, 0.9f)), (float)(lo.z + (hi.z - lo.z) * rndf(0.1f, 0.9f)));
      kind = "inside";
    }
    Vec3 dd = (aim - o).normalized();
    Vec3 d((float)dd.x, (float)dd.y, (float)dd.z);
    if (i % 12 == 11) { // parallel to a pair of faces
      const int ax = (i / 12) % 3;
      if (ax == 0) d.x = 0; else if (ax == 1) d.y = 0; else d.z = 0;
      kind = "parallel";
    }
    const bool inBox = o.x > lo.x && o.x < hi.x && o.y > lo.y && o.y < hi.y && o.z > lo.z && o.z < hi.z;
    if (inBox) kind = "inside";
    HitRecord rec{};
    const bool hit = small.vol->hit(Ray(o, d), 0.001, 1e10, rec);
    std::printf("{\"kind\": \"%s\", \"origin\": [%.17g, %.17g, %.17g], \"dir\": [%.17g, %.17g, %.17g], \"hit\": %d, \"t\": %.17g}%s\n", kind,
                o.x, o.y, o.z, d.x, d.y, d.z, hit ? 1 : 0, hit ? rec.t : 0.0, i + 1 < NH ? "," : "");
  }
  std::printf("],\n");

  // ---- Volumetric::getDensity ----
  std::printf("\"density\": [\n");
  const int ND = 400;
  for (int i = 0; i < ND; ++i) {
    // a voxel (also some outside the volume) and a point well inside it (away from the faces, where the
    // shader's float arithmetic and this double arithmetic could round to different voxels)
    const int ix = (int)(rnd32() % (uint32_t)(R[0] + 4)) - 2, iy = (int)(rnd32() % (uint32_t)(R[1] + 4)) - 2, iz = (int)(rnd32() % (uint32_t)(R[2] + 4)) - 2;
    Vec3 p((float)(pos.x + (ix + rndf(0.1f, 0.9f)) * scale), (float)(pos.y + (iy + rndf(0.1f, 0.9f)) * scale), (float)(pos.z + (iz + rndf(0.1f, 0.9f)) * scale));
    const double dens = small.vol->getDensity(p);
    std::printf("{\"point\": [%.17g, %.17g, %.17g], \"voxel\": [%d, %d, %d], \"density\": %.17g}%s\n", p.x, p.y, p.z, ix, iy, iz, dens,
                i + 1 < ND ? "," : "");
  }
  std::printf("]\n}\n");
  std::remove(datPath.c_str());
  std::remove(rawPath.c_str());
  std::remove(dir);
  return 0;
}
