# Build the MI355X hot-path library (C ABI) and the CPU oracle.
#   make            -> cuvs_amd/libcuvs_c.so  (hipcc, gfx950)  +  oracle/liboracle.so (gcc)
# Objects go to build/ (git-ignored); the .so files stay in-tree so they travel to the GPU box.
HIPCC      ?= hipcc
ARCH       ?= gfx950
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -Icuvs_amd/csrc \
              -Wno-unused-result -ffp-contract=off
# make PRODUCTION=1: the CUVS_AMD_* kernel-selection / ablation switches are compiled out (core.hip: debug_switches_on)
ifeq ($(PRODUCTION),1)
HIPFLAGS   += -DCUVS_AMD_NO_DEBUG_SWITCHES
endif
CC         ?= gcc
CFLAGS     := -O3 -march=x86-64-v3 -fPIC -fopenmp -ffp-contract=off -Wall -std=c11

SRCS := $(wildcard cuvs_amd/csrc/*.hip)
OBJS := $(patsubst cuvs_amd/csrc/%.hip,build/%.o,$(SRCS))
HDRS := $(wildcard cuvs_amd/csrc/*.hpp) $(wildcard include/cuvs/*/*.h) $(wildcard include/cuvs_amd/*.h) include/dlpack/dlpack.h

all: cuvs_amd/libcuvs_c.so oracle/liboracle.so

# the PQ scan kernel lives exactly at the 128-VGPR budget of a 1024-thread workgroup: SLP vectorisation of its
# scalar fp32 LUT arithmetic into packed pairs costs ~50 spilled registers (the codebook goes to scratch)
build/ivf_pq_search.o: HIPFLAGS += -fno-slp-vectorize
# the one-wave-per-SIMD filter: accumulators (screened by the VALU) in architectural registers, the B operands in the
# accumulation registers; fmaxf chains without canonicalisation (v_max3_f32)
build/ivf_pq_filter4.o: HIPFLAGS += -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans
# the wide filter: fmaxf chains without canonicalisation (v_max3_f32); its accumulators stay in the accumulation registers
build/ivf_pq_wide.o: HIPFLAGS += -fno-honor-nans

build/%.o: cuvs_amd/csrc/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

cuvs_amd/libcuvs_c.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle/liboracle.so: $(wildcard oracle/*.c)
	$(CC) $(CFLAGS) -shared -o $@ $^ -lm

clean:
	rm -rf build cuvs_amd/libcuvs_c.so oracle/liboracle.so

.PHONY: all clean
