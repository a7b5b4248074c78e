// wildfire_roles.hip — dispatch of the field/crew wildfire kernels (wildfire_roles.inl) to the translation unit that holds the variant.
#include "wildfire_common.h"

namespace frz_wf {

int launch_roles_group_0(const WfArgs&, int, int, int, int, hipStream_t);
int launch_roles_group_1(const WfArgs&, int, int, int, int, hipStream_t);
int launch_roles_group_2(const WfArgs&, int, int, int, int, hipStream_t);
int launch_roles_group_3(const WfArgs&, int, int, int, int, hipStream_t);
int launch_roles_group_4(const WfArgs&, int, int, int, int, hipStream_t);
int launch_roles_group_5(const WfArgs&, int, int, int, int, hipStream_t);
static_assert(FRZ_WF_ROLES_GROUPS == 6, "one declaration and one case per translation unit");

int launch_roles(const WfArgs& args, int variant, int grid, int rng, int mode, hipStream_t stream) {
    switch (variant % FRZ_WF_ROLES_GROUPS) {
        case 0: return launch_roles_group_0(args, variant, grid, rng, mode, stream);
        case 1: return launch_roles_group_1(args, variant, grid, rng, mode, stream);
        case 2: return launch_roles_group_2(args, variant, grid, rng, mode, stream);
        case 3: return launch_roles_group_3(args, variant, grid, rng, mode, stream);
        case 4: return launch_roles_group_4(args, variant, grid, rng, mode, stream);
        default: return launch_roles_group_5(args, variant, grid, rng, mode, stream);
    }
}


int roles_persist_occupancy_group_0(int);
int roles_persist_occupancy_group_1(int);
int roles_persist_occupancy_group_2(int);
int roles_persist_occupancy_group_3(int);
int roles_persist_occupancy_group_4(int);
int roles_persist_occupancy_group_5(int);

int roles_persist_occupancy(int variant) {
    switch (variant % FRZ_WF_ROLES_GROUPS) {
        case 0: return roles_persist_occupancy_group_0(variant);
        case 1: return roles_persist_occupancy_group_1(variant);
        case 2: return roles_persist_occupancy_group_2(variant);
        case 3: return roles_persist_occupancy_group_3(variant);
        case 4: return roles_persist_occupancy_group_4(variant);
        default: return roles_persist_occupancy_group_5(variant);
    }
}

}  // namespace frz_wf
