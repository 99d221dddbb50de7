#define NMFK_T double
#define NMFK_SUF f64
#include "nmfk_step_impl.h"
