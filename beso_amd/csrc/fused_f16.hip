// The fp16-operand build of the one-launch kernel (BESO_PREC_FP16): fused.hip compiled with BESO_OPERAND_F16 = 1 -- the same
// phases, layouts and packed image with v_mfma_f32_16x16x32_f16 / v_cvt_pk_f16_f32 in place of their bf16 forms, under the
// entry-point names fused_*_f16 (fused.h).  layers_kernel only: no block kernels, no training forward, no split-bf16 instances.
#define BESO_OPERAND_F16 1
#include "fused.hip"
