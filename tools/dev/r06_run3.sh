#!/bin/bash
# dev build with phase stamps in the slab kernel -> tools/dev/slab_phases.py on the headline shapes
set -x

mkdir -p gpurun_out/r06 .ab
MMD_EXTRA_HIPCC_FLAGS=-DMMD_SLSTAMPS python - <<'PY'
import os, subprocess, glob
PKG = "mm_distillnet_amd"
srcs = sorted(glob.glob(PKG + "/csrc/*.hip"))
os.makedirs(".ab/st", exist_ok=True)
flags = ["--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result", "-DMMD_NO_W16", "-DMMD_SLSTAMPS"]
objs = []
for s in srcs:
    o = ".ab/st/" + os.path.basename(s)[:-4] + ".o"
    objs.append(o)
    src_o = PKG + "/build/" + os.path.basename(s)[:-4] + ".o"
    if "pw_slab" in s or not os.path.exists(src_o):
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", s, "-o", o])
    else:
        subprocess.check_call(["cp", src_o, o])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", ".ab/libst.so", *objs])
PY
for shp in "2048 1248 208" "8192 720 120" "8192 528 88" "2048 2112 352"; do
  MMD_LIB=$PWD/.ab/libst.so python tools/dev/slab_phases.py $shp
done 2>&1 | tee gpurun_out/r06/slab_phases_v3.txt
