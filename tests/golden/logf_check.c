#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
// Candidate restatement of glibc's logf (sysdeps/ieee754/flt-32/e_logf.c, ARM optimized-routines), N=16 table.
static const struct { double invc, logc; } T[16] = {
  { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 },
  { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
  { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },
  { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
  { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 },
  { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
  { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 },
  { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
  { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 },
  { 0x1p+0, 0x0p+0 },
  { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 },
  { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
  { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 },
  { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 },
  { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },
  { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 },
};
static const double Ln2 = 0x1.62e42fefa39efp-1;
static const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
static float my_logf(float x) {
  uint32_t ix; memcpy(&ix, &x, 4);
  if (ix == 0x3f800000) return 0;
  if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
    if (ix * 2 == 0) return -INFINITY;
    if (ix == 0x7f800000) return x;
    if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return NAN;
    float s = x * 0x1p23f; memcpy(&ix, &s, 4); ix -= 23 << 23;
  }
  uint32_t tmp = ix - 0x3f330000;
  int i = (tmp >> (23 - 4)) % 16;
  int k = (int32_t)tmp >> 23;
  uint32_t iz = ix - (tmp & 0xff800000);
  double invc = T[i].invc, logc = T[i].logc;
  float zf; memcpy(&zf, &iz, 4);
  double z = (double)zf;
  double r = z * invc - 1;
  double y0 = logc + (double)k * Ln2;
  double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}
int main(void) {
  uint64_t bad = 0, n = 0;
  for (uint32_t ix = 0x00000001; ix < 0x7f800000; ix += 1) {
    float x; memcpy(&x, &ix, 4);
    float a = logf(x), b = my_logf(x);
    uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    n++;
    if (ua != ub) { if (bad < 5) printf("x=%a libm=%a mine=%a\n", x, a, b); bad++; }
  }
  printf("checked %llu floats, mismatches %llu\n", (unsigned long long)n, (unsigned long long)bad);
  return 0;
}
