// one group of kernel instantiations (see kernel_list.h)
#define SPT_INSTANTIATE_GROUP_SHADE5A 1
#include "kernel_list.h"
