/* rt_flatten.h -- host-side flattening of an RtSceneDesc into the tables of rt_device.h */
#pragma once
#include <vector>
#include <string>
#include "rt_device.h"

struct RtFlatScene {
    std::vector<RtInstr>       prog;
    std::vector<RtBox>         boxes;
    std::vector<RtAffine>      affines;
    std::vector<RtPrim>        prims;
    std::vector<double>        planes;
    std::vector<RtMaterialDev> mats;
    std::vector<RtSurfaceDev>  surfs;
    std::vector<RtTexNode>     texnodes;
    std::vector<RtClip>        clips;
    std::vector<RtLightDev>    lights;
    std::vector<RtRegion>      regions;
    std::vector<RtBvhNode>     bvh;
    std::vector<RtChain>       chains;
    std::vector<int32_t>       region_child_pc, region_always, region_lca;
    std::vector<RtBox>         region_child_boxes;     /* parallel to region_child_pc: the child's padded world box as its BVH leaf holds it */
    std::vector<RtMeshInfo>    meshes;
    std::vector<RtMeshNode>    mesh_nodes;
    std::vector<RtMeshTri>     mesh_tris;
    int  irregular_leaves;      /* leaves whose recorded affine is not their innermost enclosing affine */
    int  culled_subtrees;       /* OP_BOUND instructions emitted */
    RtFlatScene() : irregular_leaves(0), culled_subtrees(0) {}
};

/* returns 0 or an RT_ERR_* code with a message in err */
int rt_flatten_scene(const RtSceneDesc *d, RtFlatScene &out, std::string &err, bool with_bounds);
/* option push_wrappers (rt_flatten.cpp): Surface / Material wrappers above a union are pushed down to its operands (default on) */
void rt_flatten_set_push_wrappers(int on);
