/* div255_exact.c -- x / 255.0f against (x * rc) corrected once through the exact remainder, rc = 1.0f / 255.0f, for
 * every float in [0, 255] (all bit patterns 0 .. 0x437F0000, denormals included).  The volume kernel's trilinear
 * filter divides an interpolated voxel byte by 255 this way; the oracle divides.  Prints the number of mismatches. */
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
static volatile uint64_t bad[8];
static void *work(void *arg) {
  int id = (int)(intptr_t)arg;
  const float rc = 1.0f / 255.0f;
  uint64_t nb = 0;
  for (uint32_t b = id; b <= 0x437F0000u; b += 8) {
    float x; memcpy(&x, &b, 4);
    float ref = x / 255.0f;
    float q = x * rc;
    float r = fmaf(-255.0f, q, x);
    float q2 = fmaf(r, rc, q);
    if (memcmp(&ref, &q2, 4)) { if (nb < 3) printf("x=%a ref=%a got=%a\n", x, ref, q2); nb++; }
  }
  bad[id] = nb; return 0;
}
int main() { pthread_t t[8]; for (int i = 0; i < 8; ++i) pthread_create(&t[i], 0, work, (void *)(intptr_t)i);
  uint64_t s = 0; for (int i = 0; i < 8; ++i) { pthread_join(t[i], 0); s += bad[i]; } printf("mismatches %llu\n", (unsigned long long)s); return 0; }
