#!/bin/bash
# A/B builds of the library with other tuning macros (RTX_WF_TRACE_WAVES, RTX_WF_SERVICE, RTX_POOL_K, RTX_MESH_WAVES, ...):
#   tools/build_variant.sh NAME FLAGS...   -> rust-raytracing_amd/lib_variant_NAME.so (git-ignored; travels to the GPU box)
#   RTX_HIP_LIB=$PWD/rust-raytracing_amd/lib_variant_NAME.so python tools/ab_kernels.py C3 8 5 6
cd /root/repo/rust-raytracing_amd/csrc
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" \
  rtx_kernels.hip rtx_bvh.hip rtx_bvh_spheres.hip rtx_bvh_spheres_pool.hip rtx_bvh_regroup.hip rtx_bvh_mesh.hip rtx_wavefront.hip rtx_wavefront_spheres.hip rtx_api.hip \
  -o /root/repo/rust-raytracing_amd/lib_variant_$name.so
