// conv3x3_wino4h.hip -- the F(4x4,3x3) kernel of conv3x3_wino4.hip with its position products as a three-product f16 split on the matrix cores
// (nd_conv3x3_wino4h_nhwc_f32, nd_conv3x3_wino4h_16_nhwc_f32, nd_pack_conv3x3_wino4h_weight): the same source, compiled a second time with W4_F16X3 = 1.
// What differs is bracketed by `#if W4_F16X3` there: the MFMA instruction and its operand layout, the split of V when the transform writes it, the weight
// packing (U 2^11 as two f16 terms), the 2^-11 on the outputs.  Everything else -- staging, transforms, epilogue, the three wave organisations -- is shared.
#define W4_F16X3 1
#include "conv3x3_wino4.hip"
