/* Host-side AddressSanitizer driver for the C-ABI launch layer (the reference relies on PyTorch's own
 * sanitizer builds; here the launch layer is ours).  Built and run by tools/asan_host.sh against an
 * -fsanitize=address (host code only, -fno-gpu-sanitize) build of csrc/ .  Needs NO GPU: it walks the entry points'
 * host paths - version / error strings, the process-wide switches, every workspace-size query, and the argument checks
 * that must refuse a call before anything is launched (null pointers, unsupported head sizes / dtypes / shapes).  A call
 * that gets as far as the HIP runtime fails there with a launch error on a box without a device; what is checked is that
 * nothing reads or writes out of bounds on the way and that every refusal is a status code, not a crash. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "melgpt.h"

static int failures = 0;
#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);               \
      ++failures;                                                            \
    }                                                                        \
  } while (0)

int main(void) {
  EXPECT(melgpt_abi_version() == 1);
  for (int code = -8; code <= 2; ++code) {
    const char* s = melgpt_strerror(code);
    EXPECT(s != NULL && strlen(s) > 0);
  }
  /* process-wide switches */
  const int r0 = melgpt_get_reserved_cus();
  EXPECT(melgpt_set_reserved_cus(16) == MELGPT_OK && melgpt_get_reserved_cus() == 16);
  EXPECT(melgpt_set_reserved_cus(-1) != MELGPT_OK && melgpt_get_reserved_cus() == 16);
  EXPECT(melgpt_set_reserved_cus(r0) == MELGPT_OK);
  const int p0 = melgpt_get_gemm_pingpong();
  melgpt_set_gemm_pingpong(0);
  EXPECT(melgpt_get_gemm_pingpong() == 0);
  melgpt_set_gemm_pingpong(p0);
  melgpt_set_attn_bwd_two_pass(1);
  melgpt_set_attn_bwd_two_pass(0);
  /* workspace / size queries: pure host arithmetic */
  EXPECT(melgpt_vq_image_bytes(0) > 0 && melgpt_vq_image_bytes(1) > melgpt_vq_image_bytes(0));
  EXPECT(melgpt_wgrad_rowsum_rows(4096, 4) > 0);
  EXPECT(melgpt_linear_lds_workspace(64, 1024, 4096) >= 0);
  EXPECT(melgpt_layernorm_bwd_nwaves(33920) > 0 && melgpt_layernorm_bwd_nwaves(1) > 0);
  EXPECT(melgpt_colsum_rows() > 0);
  EXPECT(melgpt_groupnorm_nchunks(80 * 848) > 0 && melgpt_groupnorm_nchunks(1) > 0);
  EXPECT(melgpt_conv_in_c1_stats_workspace(4, 80, 848) > 0);
  EXPECT(melgpt_conv3x3_gn_stats_workspace(4, 80, 848) > 0);
  /* refusals: every one of these must come back as a status code */
  float f[64] = {0};
  int64_t idx[8] = {0};
  EXPECT(melgpt_gemm(NULL, 0, 64, 0, NULL, 0, 64, 0, NULL, 64, 0, 16, 16, 64, 1, MELGPT_BF16, 0, 0, 1.0f, NULL, MELGPT_ACT_NONE,
                     NULL, 0, 0, NULL, 0.f, 0, 0, NULL) != MELGPT_OK);
  EXPECT(melgpt_gemm(f, 0, 64, 0, f, 0, 64, 0, f, 64, 0, -1, 16, 64, 1, 77, 0, 0, 1.0f, NULL, MELGPT_ACT_NONE, NULL, 0, 0, NULL,
                     0.f, 0, 0, NULL) != MELGPT_OK);
  EXPECT(melgpt_attn_fwd(f, f, f, 48, f, 48, f, NULL, 1, 1, 4, 48, 0, 0.f, 1, 0, MELGPT_BF16, NULL) != MELGPT_OK);
  EXPECT(melgpt_attn_bwd(f, f, f, 48, f, f, 48, f, f, f, f, f, 48, 1, 1, 4, 48, 0, 0.f, 1, 0, MELGPT_BF16, NULL) != MELGPT_OK);
  EXPECT(melgpt_cast(f, 99, f, MELGPT_BF16, 8, NULL) != MELGPT_OK);
  EXPECT(melgpt_vq_gather(idx, 4, f, 128, 7, f, MELGPT_F32, 1, 7, 1, 1, NULL) != MELGPT_OK);
  EXPECT(melgpt_vq_gather(NULL, 4, f, 128, 256, f, MELGPT_F32, 1, 256, 1, 1, NULL) != MELGPT_OK);
  if (failures == 0) printf("asan host driver: ok\n");
  return failures ? 1 : 0;
}
