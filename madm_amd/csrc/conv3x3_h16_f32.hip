// The f32 instantiation of the 16 x 16-pixel halo conv (conv3x3_h16.hip), built without packed-FP32 VALU ops (Makefile: NOPK).
#define H16_ONLY_F32
#include "conv3x3_h16.hip"
