// sc_tick_hostmath.cpp -- host-side mirror of the four sc_math functions the camera path uses
// (src/core/src/sc_math.cpp), exported through the C ABI.  Plain C++, no GPU work.
// Compiled with -ffp-contract=off: the reference is /fp:precise.
#include "../../include/sc_tick.h"

#include <cmath>
#include <cstring>

namespace {

struct M4 { float m[16]; };

M4 ident()
{
  M4 r; std::memset(r.m, 0, sizeof r.m);
  r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
  return r;
}

// column k of a scaled by b[col][k], summed left to right from the first product (sc_math.cpp:52-68)
M4 mul(const M4& a, const M4& b)
{
  M4 r;
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 4; ++row) {
      float s = a.m[row] * b.m[col * 4];
      for (int k = 1; k < 4; ++k) s = s + a.m[k * 4 + row] * b.m[col * 4 + k];
      r.m[col * 4 + row] = s;
    }
  return r;
}

M4 axisRotation(int axis, float angle)
{
  // plane of rotation (u, v): x -> (y,z), y -> (z,x), z -> (x,y); R[u,u]=c R[v,u]=s R[u,v]=-s R[v,v]=c
  const int u = (axis + 1) % 3, v = (axis + 2) % 3;
  const float c = std::cos(angle), s = std::sin(angle);
  M4 r = ident();
  r.m[u * 4 + u] = c; r.m[u * 4 + v] = s; r.m[v * 4 + u] = -s; r.m[v * 4 + v] = c;
  return r;
}

M4 trs(const float* pos, const float* rot, const float* scale)
{
  M4 t = ident();
  t.m[12] = pos[0]; t.m[13] = pos[1]; t.m[14] = pos[2];
  const M4 rx = axisRotation(0, rot[0]), ry = axisRotation(1, rot[1]), rz = axisRotation(2, rot[2]);
  const M4 r = mul(mul(rz, ry), rx);                       // sc_math.cpp:127
  M4 s; std::memset(s.m, 0, sizeof s.m);
  s.m[0] = scale[0]; s.m[5] = scale[1]; s.m[10] = scale[2]; s.m[15] = 1.0f;
  return mul(t, mul(r, s));                                // sc_math.cpp:141
}

// Adjugate by 3x3 minors would round differently from the reference; its expansion (sc_math.cpp:150-196)
// is a fixed list of signed triple products per entry, encoded here as index triples in its order.
const signed char kAdj[16][19] = {
  /* slot, then 6 x (sign*(i+1), j, k) with sign folded into the first index */
  { 0,  +6,10,15,  -6,11,14, -10, 6,15, +10, 7,14, +14, 6,11, -14, 7,10},
  { 4,  -5,10,15,  +5,11,14,  +9, 6,15,  -9, 7,14, -13, 6,11, +13, 7,10},
  { 8,  +5, 9,15,  -5,11,13,  -9, 5,15,  +9, 7,13, +13, 5,11, -13, 7, 9},
  {12,  -5, 9,14,  +5,10,13,  +9, 5,14,  -9, 6,13, -13, 5,10, +13, 6, 9},
  { 1,  -2,10,15,  +2,11,14, +10, 2,15, -10, 3,14, -14, 2,11, +14, 3,10},
  { 5,  +1,10,15,  -1,11,14,  -9, 2,15,  +9, 3,14, +13, 2,11, -13, 3,10},
  { 9,  -1, 9,15,  +1,11,13,  +9, 1,15,  -9, 3,13, -13, 1,11, +13, 3, 9},
  {13,  +1, 9,14,  -1,10,13,  -9, 1,14,  +9, 2,13, +13, 1,10, -13, 2, 9},
  { 2,  +2, 6,15,  -2, 7,14,  -6, 2,15,  +6, 3,14, +14, 2, 7, -14, 3, 6},
  { 6,  -1, 6,15,  +1, 7,14,  +5, 2,15,  -5, 3,14, -13, 2, 7, +13, 3, 6},
  {10,  +1, 5,15,  -1, 7,13,  -5, 1,15,  +5, 3,13, +13, 1, 7, -13, 3, 5},
  {14,  -1, 5,14,  +1, 6,13,  +5, 1,14,  -5, 2,13, -13, 1, 6, +13, 2, 5},
  { 3,  -2, 6,11,  +2, 7,10,  +6, 2,11,  -6, 3,10, -10, 2, 7, +10, 3, 6},
  { 7,  +1, 6,11,  -1, 7,10,  -5, 2,11,  +5, 3,10,  +9, 2, 7,  -9, 3, 6},
  {11,  -1, 5,11,  +1, 7, 9,  +5, 1,11,  -5, 3, 9,  -9, 1, 7,  +9, 3, 5},
  {15,  +1, 5,10,  -1, 6, 9,  -5, 1,10,  +5, 2, 9,  +9, 1, 6,  -9, 2, 5},
};

M4 inverse(const M4& a)
{
  const float* m = a.m;
  float o[16];
  for (int e = 0; e < 16; ++e) {
    const signed char* t = kAdj[e];
    float acc = 0.0f;
    for (int q = 0; q < 6; ++q) {
      const int si = t[1 + 3 * q];
      const int i = (si < 0 ? -si : si) - 1, j = t[2 + 3 * q], k = t[3 + 3 * q];
      const float p = (m[i] * m[j]) * m[k];
      if (q == 0) acc = si < 0 ? -p : p;
      else acc = si < 0 ? acc - p : acc + p;
    }
    o[t[0]] = acc;
  }
  const float det = m[0] * o[0] + m[1] * o[4] + m[2] * o[8] + m[3] * o[12];
  if (std::fabs(det) <= 1e-6f) return ident();           // EPSILON, sc_math.h:6
  const float inv = 1.0f / det;
  M4 r;
  for (int i = 0; i < 16; ++i) r.m[i] = o[i] * inv;
  return r;
}

M4 perspective(float fov, float aspect, float zn, float zf, bool flipY)
{
  const float eps = 1e-6f;
  if (fov <= eps || aspect <= eps || zn <= eps || zf <= zn + eps) return ident();
  M4 r; std::memset(r.m, 0, sizeof r.m);
  const float f = 1.0f / std::tan(fov * 0.5f);
  r.m[0] = f / aspect;
  r.m[5] = flipY ? -f : f;
  r.m[10] = zf / (zn - zf);
  r.m[14] = (zf * zn) / (zn - zf);
  r.m[11] = -1.0f;
  return r;
}

M4 load(const float* p) { M4 r; std::memcpy(r.m, p, 64); return r; }

} // namespace

extern "C" {

int scTickHostMat4Mul(const float a[16], const float b[16], float out[16])
{
  if (!a || !b || !out) return 0;
  const M4 r = mul(load(a), load(b));
  std::memcpy(out, r.m, 64);
  return 1;
}

int scTickHostMat4Trs(const float pos[3], const float rot[3], const float scale[3], float out[16])
{
  if (!out) return 0;
  const M4 r = (pos && rot && scale) ? trs(pos, rot, scale) : ident();   // sc_math.cpp:135-136
  std::memcpy(out, r.m, 64);
  return 1;
}

int scTickHostMat4Inverse(const float a[16], float out[16])
{
  if (!a || !out) return 0;
  const M4 r = inverse(load(a));
  std::memcpy(out, r.m, 64);
  return 1;
}

int scTickHostMat4PerspectiveRhZo(float fov, float aspect, float zn, float zf, int flipY, float out[16])
{
  if (!out) return 0;
  const M4 r = perspective(fov, aspect, zn, zf, flipY != 0);
  std::memcpy(out, r.m, 64);
  return 1;
}

int scTickHostCameraViewProj(const float camWorld[16], float fovYDeg, float aspect, float zn, float zf, float out[16])
{
  if (!camWorld || !out) return 0;
  const float fovRad = fovYDeg * 3.1415926535f / 180.0f;             // sc_ecs.cpp:263
  const M4 proj = perspective(fovRad, aspect, zn, zf, true);
  const M4 view = inverse(load(camWorld));
  const M4 vp = mul(proj, view);
  std::memcpy(out, vp.m, 64);
  return 1;
}

} // extern "C"
