#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static inline double div3(double x, double c, double rc) {
  double q0 = x * rc;
  double r = fma(-c, q0, x);
  return fma(r, rc, q0);
}
static uint64_t s[4] = {1, 2, 3, 4};
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t nxt(void) { uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
int main(void) {
  volatile double cv = .0001;
  const double c = cv, rc = 1.0 / c;
  long bad = 0, n = 0;
  // boundaries: x near k*c
  for (int k = 0; k <= 100001; ++k) {
    double x0 = k * c;
    for (int u = -40; u <= 40; ++u) {
      double x = x0;
      for (int a = 0; a < (u < 0 ? -u : u); ++a) x = nextafter(x, u < 0 ? -1.0 : 20.0);
      if (x < 0) continue;
      volatile double q = x / c;
      if (div3(x, c, rc) != q) { if (bad < 5) printf("mismatch x=%.17g\n", x); ++bad; }
      ++n;
    }
  }
  // dx / c for dx in [0, 1e-4]
  for (long i = 0; i < 400000000L; ++i) {
    uint64_t r = nxt();
    double x = (double)(r >> 11) * (1.0 / 9007199254740992.0);
    if (i & 1) x *= 10.0; else x *= 1.0001e-4;
    volatile double q = x / c;
    if (div3(x, c, rc) != q) { if (bad < 5) printf("mismatch x=%.17g\n", x); ++bad; }
    ++n;
  }
  printf("tested %ld, mismatches %ld, rc=%.17g\n", n, bad, rc);
  return 0;
}
