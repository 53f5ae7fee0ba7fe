// The per-lane walk kernels as their own translation unit: the same source as rt_device.hip, built
// with -fno-slp-vectorize (see the note at the top of that file).
#define RT_TU_WALK 1
#include "rt_device.hip"
