# Builds, in-tree:
#   simple-path-tracer_amd/lib/libspt_host.so  host side (scene JSON/OBJ/EXR -> spt_scene_desc, PNG)
#   simple-path-tracer_amd/lib/libspt_hip.so   HIP kernels + C ABI (gfx950)
#   simple-path-tracer_amd/lib/libspt_hip_bez.so  the same with the Bezier-patch primitive (opened by libspt_hip.so on demand)
#   simple-path-tracer_amd/lib/spt             CLI with the reference's flags (src/main.rs:26-41)
#   oracle/liboracle.so                        CPU oracle (test infrastructure only)
# FP contraction is disabled everywhere so the deterministic math of
# include/spt_detmath.h gives identical bits on x86-64 and gfx950.
PKG      := simple-path-tracer_amd
LIBDIR   := $(PKG)/lib
CXX      ?= g++
HIPCC    ?= hipcc
CXXFLAGS := -std=c++17 -O2 -fPIC -Wall -Wextra -ffp-contract=off -fno-fast-math -Iinclude
# -fno-slp-vectorize: the SLP vectoriser pairs scalar f32 operations into v_pk_mul_f32 / v_pk_add_f32.  On gfx950 a packed
# f32 instruction issues in 4 cycles per wave64 (profiles/r03_issue_rate.md) - no better than two 2-cycle VOP2s - and every
# pair costs v_mov / v_pk_mov shuffles to line its operands up.  MEASURED (round 3): cfg2 81.3 -> 88.8 Gsamples/s, cfg4
# 91.9 -> 88.2 ms, cfg5 143.3 -> 138.6 ms, same bits (packed and scalar IEEE operations round alike).
HIPFLAGS := -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
            -Wall -Wno-unused-function -Iinclude

HOST_SRC := $(wildcard $(PKG)/csrc/host/*.cpp)
HOST_HDR := $(wildcard $(PKG)/csrc/host/*.hpp) $(wildcard include/*.h)
HIP_SRC  := $(wildcard $(PKG)/csrc/hip/*.hip)
HIP_HDR  := $(wildcard $(PKG)/csrc/hip/*.h) $(wildcard include/*.h)

.PHONY: all host hip hip-plain cli oracle clean
all: host oracle hip cli

host: $(LIBDIR)/libspt_host.so
hip: $(LIBDIR)/libspt_hip.so $(LIBDIR)/libspt_hip_bez.so
hip-plain: $(LIBDIR)/libspt_hip.so   # kernel iteration: the library every scene without Bezier patches uses
cli: $(LIBDIR)/spt
oracle:
	$(MAKE) -C oracle

$(LIBDIR)/libspt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -lz -lpthread

# one object per translation unit (spt_hip.hip = host side + film kernels, inst_*.hip = groups of kernel instantiations,
# see csrc/hip/kernel_list.h), compiled side by side with `make -jN`
OBJDIR   := build/hip
HIP_OBJ     := $(patsubst $(PKG)/csrc/hip/%.hip,$(OBJDIR)/plain/%.o,$(HIP_SRC))
HIP_OBJ_BEZ := $(patsubst $(PKG)/csrc/hip/%.hip,$(OBJDIR)/bez/%.o,$(HIP_SRC))

$(OBJDIR)/plain/%.o: $(PKG)/csrc/hip/%.hip $(HIP_HDR)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

# the same sources with the CubicBezier primitive compiled in; the library is opened by libspt_hip.so for scenes with patches
$(OBJDIR)/bez/%.o: $(PKG)/csrc/hip/%.hip $(HIP_HDR)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -DSPT_WITH_BEZIER=1 -c -o $@ $<

$(LIBDIR)/libspt_hip.so: $(HIP_OBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -Wl,-Bsymbolic -o $@ $(HIP_OBJ) -ldl

$(LIBDIR)/libspt_hip_bez.so: $(HIP_OBJ_BEZ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -Wl,-Bsymbolic -o $@ $(HIP_OBJ_BEZ) -ldl

$(LIBDIR)/spt: $(PKG)/csrc/cli/main.cpp $(wildcard include/*.h) $(LIBDIR)/libspt_host.so $(LIBDIR)/libspt_hip.so
	$(CXX) $(CXXFLAGS) -o $@ $< -L$(LIBDIR) -lspt_host -lspt_hip -Wl,-rpath,'$$ORIGIN'

clean:
	rm -rf $(LIBDIR) build oracle/*.so oracle/_ref
