#define NMFK_T float
#define NMFK_SUF f32
#include "nmfk_step_impl.h"
