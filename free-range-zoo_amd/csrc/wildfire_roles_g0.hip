// one of the FRZ_WF_ROLES_GROUPS translation units of the field/crew wildfire kernels (wildfire_roles.inl): variants i with i % groups == 0
#define FRZ_WF_ROLES_GROUP 0
#include "wildfire_roles.inl"
