// G1 instantiation of the MSM pipeline (see msm_impl.inc)
#define VSP_MSM_GROUP 1
#include "msm_impl.inc"
