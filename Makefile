# Top-level build.  Everything is built IN-TREE so the binaries travel to the GPU box
# with the source snapshot (they are git-ignored, not gpurun-ignored).
#
#   make            shim + its PT_DIAG twin + its development build + host library + CLI + oracle checkers
#   make shim       raytracer.c_amd/csrc/librt_hip.so        (hipcc, gfx950 only)
#   make host       raytracer.c_amd/host/libraytracer_amd.so + raytracer (gcc, C99)
#   make oracle     oracle/libpt_oracle.so (+ oracle/_ref/*.so when /root/reference exists)

ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
PKG     := $(ROOT)raytracer.c_amd
CSRC    := $(PKG)/csrc
HOST    := $(PKG)/host
INC     := $(ROOT)include

HIPCC   ?= /opt/rocm/bin/hipcc
# -ffp-contract=off: decision-exact parity with the reference (gcc --std=c99 emits no FMA);
# hipcc's default would fuse a*b+c.  No fast-math: IEEE fp64 sqrt and division.
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function \
            -I$(INC) -I$(CSRC)
CC      := gcc
CFLAGS  := -std=c99 -D_DEFAULT_SOURCE -O2 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I$(INC) -I$(HOST)

SHIM    := $(CSRC)/librt_hip.so
# the device code is ONE translation unit, pt_kernel.hip, split by topic into the pt_*.h headers it includes
KERNEL_SRC := $(CSRC)/pt_kernel.hip $(CSRC)/rt_hip_shim.hip $(wildcard $(CSRC)/pt_*.h) $(CSRC)/bvh_build.h $(INC)/rt_hip.h $(INC)/rt_rng.h
HOSTLIB := $(HOST)/libraytracer_amd.so
CLI     := $(HOST)/raytracer

HOST_SRC := $(HOST)/raytracer_amd.c $(HOST)/scenes.c $(HOST)/obj_load.c $(HOST)/png_out.c
HOST_HDR := $(INC)/raytracer.h $(INC)/vector.h $(INC)/rt_hip.h $(INC)/rt_rng.h $(HOST)/scenes.h

all: shim shim-diag shim-dev host oracle

# oracle/_ref/ref_main_dropin links against the host library: build that first
oracle: host

shim: $(SHIM)
$(SHIM): $(KERNEL_SRC)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/pt_kernel.hip $(CSRC)/rt_hip_shim.hip -lrccl

# diagnostic build: wave-level event counters in stats[4..] and the exhaustive re-checks of every conservative rule
# (filter, fp32 pre-tests, bounding-sphere probe, hull facets).  Not the product: only tests/test_gpu_diag.py
# (child processes with RT_HIP_SHIM_PATH) and tools/diag*.py load it.  Built by `make all` so that it travels to
# the GPU box with the snapshot like the other binaries (git-ignored, not gpurun-ignored).
shim-diag: $(CSRC)/librt_hip_diag.so
$(CSRC)/librt_hip_diag.so: $(KERNEL_SRC)
	$(HIPCC) $(HIPFLAGS) -DPT_DIAG -shared -o $@ $(CSRC)/pt_kernel.hip $(CSRC)/rt_hip_shim.hip -lrccl

# development build (-DPT_DEV_KERNELS): the product plus what only A/B runs and tests of the fallbacks need -- pt_render_tiles_v0
# (the literal single-phase scan), RT_HIP_KERNEL_VARIANT (other arms of the pick table), RT_HIP_FORCE_BIG, RT_HIP_NO_BIG_PRUNE,
# RT_HIP_EXTRA_LDS, RT_HIP_BVH_MEDIAN, RT_HIP_POOL_SLOTS (pools of n slots per XCD: n = 1 makes slot acquisition FAIL, which is
# how the status-word path is tested).  None of these switches exists in librt_hip.so.  Loaded through RT_HIP_SHIM_PATH by
# child processes of the tests and by tools/; built by `make all` so that it travels to the GPU box with the snapshot.
shim-dev: $(CSRC)/librt_hip_dev.so
$(CSRC)/librt_hip_dev.so: $(KERNEL_SRC)
	$(HIPCC) $(HIPFLAGS) -DPT_DEV_KERNELS -shared -o $@ $(CSRC)/pt_kernel.hip $(CSRC)/rt_hip_shim.hip -lrccl

host: $(HOSTLIB) $(CLI)
$(HOSTLIB): $(HOST_SRC) $(HOST_HDR) $(SHIM)
	$(CC) $(CFLAGS) -shared -o $@ $(HOST_SRC) -L$(CSRC) -lrt_hip -Wl,-rpath,'$$ORIGIN/../csrc' -lz -lm
$(CLI): $(HOST)/main.c $(HOSTLIB)
	$(CC) $(CFLAGS) -o $@ $(HOST)/main.c -L$(HOST) -lraytracer_amd -L$(CSRC) -lrt_hip -Wl,-rpath,'$$ORIGIN' \
	    -Wl,-rpath,'$$ORIGIN/../csrc' -lm

oracle:
	$(MAKE) -C $(ROOT)oracle all

clean:
	rm -f $(SHIM) $(CSRC)/librt_hip_diag.so $(CSRC)/librt_hip_dev.so $(HOSTLIB) $(CLI)
	$(MAKE) -C $(ROOT)oracle clean

.PHONY: all shim shim-diag shim-dev host oracle clean

# development: alternative builds of the shim for A/B runs (tools/gpu_ab.py), e.g.
#   make variant NAME=tri4 DEFS="-DPT_MIN_WAVES_TRI=4"   ->  raytracer.c_amd/csrc/variants/librt_hip_tri4.so
variant:
	@mkdir -p $(CSRC)/variants
	$(HIPCC) $(HIPFLAGS) $(DEFS) -shared -o $(CSRC)/variants/librt_hip_$(NAME).so $(CSRC)/pt_kernel.hip $(CSRC)/rt_hip_shim.hip -lrccl
.PHONY: variant
