// G1 instantiation of the generator-side batch exponentiation (see fixedbase_impl.inc)
#define VSP_FB_GROUP 1
#include "fixedbase_impl.inc"
