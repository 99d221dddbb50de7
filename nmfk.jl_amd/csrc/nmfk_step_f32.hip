#define NMFK_T float
#define NMFK_SUF f32
#define NMFK_IS_F32 1
#include "nmfk_step_impl.h"
