// ref_ray_driver.cpp -- runs ray/box cases through the real sc::editor::intersectRayAABB (oracle/_ref/libsc_ref_editor.so).
// TEST INFRASTRUCTURE ONLY; used by oracle/make_golden_rays.py where /root/reference exists.
//   stdin : N records of 12 float32 (origin xyz, dir xyz, box min xyz, box max xyz)
//   stdout: N records of (int32 hit, float32 t)   -- t = 0 when the reference did not write it (a miss)
// The library is opened with RTLD_LAZY: it holds undefined scRender* references (the editor's renderer calls) that this
// path never reaches.
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char** argv)
{
  if (argc < 2) { std::fprintf(stderr, "usage: ref_ray_driver <libsc_ref_editor.so>\n"); return 2; }
  void* lib = dlopen(argv[1], RTLD_LAZY | RTLD_LOCAL);
  if (!lib) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  using Fn = int (*)(const float*, const float*, const float*, const float*, float*);
  Fn fn = reinterpret_cast<Fn>(dlsym(lib, "ref_intersect_ray_aabb"));
  if (!fn) { std::fprintf(stderr, "dlsym: %s\n", dlerror()); return 4; }
  std::vector<float> in;
  float buf[12 * 256];
  size_t got;
  while ((got = std::fread(buf, sizeof(float), 12 * 256, stdin)) > 0) in.insert(in.end(), buf, buf + got);
  if (in.size() % 12) { std::fprintf(stderr, "input is not a whole number of 12-float records\n"); return 5; }
  for (size_t i = 0; i < in.size(); i += 12) {
    float t = 0.0f;
    const int32_t hit = fn(&in[i], &in[i + 3], &in[i + 6], &in[i + 9], &t);
    std::fwrite(&hit, sizeof hit, 1, stdout);
    std::fwrite(&t, sizeof t, 1, stdout);
  }
  return 0;
}
