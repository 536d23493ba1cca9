#!/bin/bash
# A/B builds of the library with other tuning macros (RTX_WF_TRACE_WAVES, RTX_WF_SERVICE, RTX_POOL_K, RTX_MESH_WAVES, RTX_SPK_WAVES, ...):
#   tools/build_variant.sh NAME FLAGS...   -> rust-raytracing_amd/lib_variant_NAME.so (git-ignored; travels to the GPU box)
#   (LAB=1 tools/build_variant.sh ...: a variant of the lab library)
#   RTX_HIP_LIB=$PWD/rust-raytracing_amd/lib_variant_NAME.so python tools/ab_kernels.py C3 8 5 6
name=$1; shift
python3 - "$name" "$@" <<'PY'
import importlib.util, sys
spec = importlib.util.spec_from_file_location("b", "/root/repo/rust-raytracing_amd/build.py")
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import os
print(b.build(extra_flags=sys.argv[2:], lib="/root/repo/rust-raytracing_amd/lib_variant_%s.so" % sys.argv[1], lab=bool(os.environ.get("LAB"))))
PY
