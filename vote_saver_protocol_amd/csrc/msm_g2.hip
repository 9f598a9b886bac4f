// G2 instantiation of the MSM pipeline (see msm_impl.inc)
#define VSP_MSM_GROUP 2
#include "msm_impl.inc"
