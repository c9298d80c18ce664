// Experiment (r5): conv3x3_wino4x.hip = conv3x3_wino4.hip as of branch exp/r5-bf16x3 (the F(4x4) kernel with a W4_BF16X3 form: position products on
// v_mfma_f32_16x16x32_bf16, every operand as three bf16 terms, six products).  Built only by tools/experiments/wino4b/build.sh into tools/_build/.
#define W4_BF16X3 1
#include "conv3x3_wino4x.hip"
