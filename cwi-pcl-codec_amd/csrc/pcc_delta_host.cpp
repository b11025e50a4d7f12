// pcc_delta_host.cpp -- RigidTransformCoding / QuaternionCoding of the reference, restated
// (impl/rigid_transform_coding_impl.hpp:63-203, impl/quaternion_coding_impl.hpp:55-222).  Float arithmetic in the
// reference's order; Eigen::Quaternion<float>(Matrix3f) and toRotationMatrix() as Eigen 3.3 computes them.
#include <float.h>
#include <math.h>
#include <string.h>

#include "pcc_delta.h"

namespace pcc {
namespace {

struct Quat { float x, y, z, w; };

Quat quat_from_matrix(const float m[3][3]) {  // Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Matrix3>
  Quat q;
  float* c = &q.x;  // coeffs: x y z w
  float t = (m[0][0] + m[1][1]) + m[2][2];
  if (t > 0.0f) {
    t = sqrtf(t + 1.0f);
    q.w = 0.5f * t;
    t = 0.5f / t;
    q.x = (m[2][1] - m[1][2]) * t;
    q.y = (m[0][2] - m[2][0]) * t;
    q.z = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(((m[i][i] - m[j][j]) - m[k][k]) + 1.0f);
    c[i] = 0.5f * t;
    t = 0.5f / t;
    q.w = (m[k][j] - m[j][k]) * t;
    c[j] = (m[j][i] + m[i][j]) * t;
    c[k] = (m[k][i] + m[i][k]) * t;
  }
  return q;
}

void quat_to_matrix(const Quat& q, float m[3][3]) {  // QuaternionBase::toRotationMatrix
  const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  m[0][0] = 1.0f - (tyy + tzz); m[0][1] = txy - twz; m[0][2] = txz + twy;
  m[1][0] = txy + twz; m[1][1] = 1.0f - (txx + tzz); m[1][2] = tyz - twx;
  m[2][0] = txz - twy; m[2][1] = tyz + twx; m[2][2] = 1.0f - (txx + tyy);
}

inline float clamp1(float v) { return v < -1 ? -1.0f : (v > 1 ? 1.0f : v); }

// compressQuaternion (quaternion_coding_impl.hpp:55-166)
void quat_compress(const Quat& q, int16_t s[3]) {
  static const float scale = 1.41421f;
  const float x = q.x, y = q.y, z = q.z, w = q.w;
  auto pack = [&](float a, float b, float c, bool neg, int bit1, int bit2) {
    float ra = a * scale, rb = b * scale, rc = c * scale;
    if (neg) { ra = -ra; rb = -rb; rc = -rc; }
    ra = clamp1(ra); rb = clamp1(rb); rc = clamp1(rc);
    s[0] = int16_t(ra * 32767);
    s[1] = int16_t((int(rb * 32767) & 0xfffe) | bit1);
    s[2] = int16_t((int(rc * 32767) & 0xfffe) | bit2);
  };
  if (w > x && w > y && w > z) pack(x, y, z, w < 0, 1, 1);
  else if (z > x && z > y) pack(x, y, w, z < 0, 1, 0);
  else if (y > x) pack(x, z, w, y < 0, 0, 1);
  else pack(y, z, w, x < 0, 0, 0);
}

// deCompressQuaternion (:168-222); clears the low bits of s[1], s[2] in place like the reference
Quat quat_decompress(int16_t s[3]) {
  const int which = ((s[1] & 1) << 1) | (s[2] & 1);
  s[1] &= (int16_t)0xfffe;
  s[2] &= (int16_t)0xfffe;
  static const float scale = 1.0f / 32767.0f / 1.41421f;
  const float a = s[0] * scale, b = s[1] * scale, c = s[2] * scale;
  float d = 1 - (a * a) - (b * b) - (c * c);
  if (d > FLT_EPSILON) d = sqrtf(d);
  Quat q;
  if (which == 3) { q.x = a; q.y = b; q.z = c; q.w = d; }
  else if (which == 2) { q.x = a; q.y = b; q.w = c; q.z = d; }
  else if (which == 1) { q.x = a; q.z = b; q.w = c; q.y = d; }
  else { q.y = a; q.z = b; q.w = c; q.x = d; }
  return q;
}

}  // namespace

void rigid_compress(const float tr[16], std::vector<int16_t>& comp) {  // rigid_transform_coding_impl.hpp:63-149
  const float scaling_factor = (float)32767 / 2.5;
  float rot[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) rot[r][c] = tr[4 * r + c];
  const Quat q = quat_from_matrix(rot);
  comp.assign(3, 0);
  quat_compress(q, comp.data());
  float res_rot[3][3];
  quat_to_matrix(quat_decompress(comp.data()), res_rot);
  bool stable = true;
  for (int i = 0; i < 9; ++i) {
    if (fabsf(res_rot[i / 3][i % 3] - tr[4 * (i / 3) + i % 3]) > 0.001) { stable = false; break; }
  }
  if (!stable) {
    comp.resize(7);
    comp[3] = comp[4] = comp[5] = comp[6] = 0;
    for (int l = 0; l < 3; ++l) {
      comp[l] = (int16_t)int(rot[0][l] * (32767 - 1));
      comp[l + 3] = (int16_t)int(rot[1][l] * (32767 - 1));
      comp[6] += rot[2][l] < 0 ? 1 << l : 0;
    }
  } else {
    quat_compress(q, comp.data());
  }
  for (int a = 0; a < 3; ++a) {
    float t = tr[4 * a + 3];
    if (t > 2.5) t = 2.5;
    if (t < -2.5) t = -2.5;
    comp.push_back((int16_t)(int)(t * (scaling_factor - 1)));
  }
}

void rigid_decompress(const int16_t* comp_in, size_t count, float tr[16]) {  // :158-203
  const float scaling_factor = (float)32767 / 2.5;
  for (int i = 0; i < 16; ++i) tr[i] = 0.0f;  // (the reference starts from what the caller passes: identity)
  tr[0] = tr[5] = tr[10] = 1.0f;
  if (count == 6) {
    int16_t s[3] = {comp_in[0], comp_in[1], comp_in[2]};
    float m[3][3];
    quat_to_matrix(quat_decompress(s), m);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) tr[4 * r + c] = m[r][c];
  } else {
    for (int l = 0; l < 3; ++l) {
      tr[l] = ((float)comp_in[l]) / (32767 - 1);
      tr[4 + l] = ((float)comp_in[l + 3]) / (32767 - 1);
      tr[8 + l] = sqrtf(1 - tr[l] * tr[l] - tr[4 + l] * tr[4 + l]);
      if (((1 << l) & ((int)comp_in[6])) == 1 << l) tr[8 + l] = -tr[8 + l];
    }
  }
  tr[3] = ((float)comp_in[count - 3]) / ((float)(scaling_factor - 1));
  tr[7] = ((float)comp_in[count - 2]) / ((float)(scaling_factor - 1));
  tr[11] = ((float)comp_in[count - 1]) / ((float)(scaling_factor - 1));
  tr[12] = tr[13] = tr[14] = 0;
  tr[15] = 1;
}

}  // namespace pcc
