/* Sanitizer driver for the CPU oracle: exercises map generation, the agent, the turn engine, the
 * auto-reset pool and the experience side channel under -fsanitize=address,undefined.
 * Build/run: make -C oracle sanitize   (test infrastructure; not part of the product) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "generals_oracle.h"

int main(void) {
  const int B = 96, MW = 20, MH = 20, MP = 4;
  const int sizes[3][3] = {{10, 10, 2}, {15, 15, 3}, {20, 20, 4}};
  int stride = MW * MH;   /* B * stride >= 25 * 25: the Go-generator block below reuses the first board's planes */
  int32_t* army = calloc((size_t)B * stride, 4);
  int8_t* owner = malloc((size_t)B * stride);
  uint8_t* type = calloc((size_t)B * stride, 1);
  int32_t w[96], h[96], p[96];
  memset(owner, 0xFF, (size_t)B * stride);
  for (int i = 0; i < B; i++) {
    w[i] = sizes[i % 3][0]; h[i] = sizes[i % 3][1]; p[i] = sizes[i % 3][2];
    if (ora_mapgen(7, i, w[i], h[i], p[i], army + (size_t)i * stride, owner + (size_t)i * stride, type + (size_t)i * stride)) return 2;
  }
  ora_params prm; ora_params_default(&prm);
  ora_batch* b = ora_batch_new(B, MW, MH, MP, &prm);
  if (ora_batch_reset(b, NULL, B, army, owner, type, w, h, p)) return 3;
  ora_batch_set_pool(b, 11, 5, NULL, NULL, NULL);
  ora_action8* acts = calloc((size_t)B * MP, sizeof(ora_action8));
  int32_t* err = calloc(B, 4);
  int mask_bytes = ora_mask_bytes(stride);
  uint8_t* bits = malloc((size_t)B * MP * mask_bytes);
  float* rew = malloc(sizeof(float) * B * MP);
  float* obs = malloc(sizeof(float) * (size_t)B * 9 * stride);
  uint8_t* done = malloc(B);
  long long errs = 0;
  for (int k = 0; k < 300; k++) {
    ora_batch_agent_actions(b, 3, 20, acts, 1);
    ora_batch_experience_begin(b);
    ora_batch_step(b, acts, err, bits, 1);
    ora_batch_rewards(b, rew, done);
    if (k % 50 == 0) { ora_batch_observe(b, k % MP, obs); ora_batch_serializer_mask(b, bits); }
    for (int i = 0; i < B; i++) errs += err[i] != 0;
  }
  long long steps = ora_batch_rollout(b, 100, 9, 5, 1);
  steps += ora_batch_rollout_masks(b, 60, 10, 5, 1, bits);   /* bits: [B][MP][ora_mask_bytes(stride)] */
  /* Go's math/rand and the generator on it: default and custom configs, awkward seeds */
  {
    const int64_t seeds[] = {0, 1, -3, 12345, 2147483647LL, 2147483648LL, (int64_t)1 << 40};
    const int32_t custom[7] = {10, 4, 8, 30, 35, 6, 7};
    for (unsigned i = 0; i < sizeof seeds / sizeof seeds[0]; i++) {
      if (ora_mapgen_go(seeds[i], 20, 20, 4, NULL, army, owner, type)) return 4;
      if (ora_mapgen_go(seeds[i], 25, 25, 4, custom, army, owner, type)) return 6;
      ora_gorand* r = ora_gorand_new(seeds[i]);
      long long acc = 0;
      for (int k = 0; k < 2000; k++) acc += ora_gorand_intn(r, 1 + (k % 97)) + (ora_gorand_int63(r) & 1);
      ora_gorand_free(r);
      if (acc < 0) return 5;
    }
  }
  printf("oracle sanitizer run OK: %lld aborted turns, %lld rollout steps\n", errs, steps);
  ora_batch_free(b);
  free(army); free(owner); free(type); free(acts); free(err); free(bits); free(rew); free(obs); free(done);
  return 0;
}
