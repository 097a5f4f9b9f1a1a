# Builds the HIP library (C-ABI: include/phylomap_hip.h) and the CPU oracle.
# -ffp-contract=off is part of the arithmetic spec (no fused multiply-add on either side unless written as one).
# One object per source so that `make -j` compiles the kernels side by side.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := phylomap_amd/csrc
LIB     := phylomap_amd/libphylomap_hip.so
OBJDIR  := build/obj
SRCS    := $(CSRC)/phm_engine.cpp $(CSRC)/phm_drivers.cpp $(CSRC)/phm_expm_api.cpp $(CSRC)/phm_sched.cpp $(CSRC)/phm_qupdate.cpp $(CSRC)/phm_rtc.cpp $(CSRC)/phm_mcmc.hip $(CSRC)/phm_wide.hip $(CSRC)/phm_narrow.hip $(CSRC)/phm_tiles.hip $(CSRC)/phm_wbranch.hip $(CSRC)/phm_wtiles.hip $(CSRC)/phm_exp.hip
OBJS    := $(patsubst $(CSRC)/%,$(OBJDIR)/%.o,$(SRCS))
HDRS    := $(wildcard $(CSRC)/*.h) include/phylomap_hip.h
# EXTRA: experiment switches (e.g. make LIB=scratch/libv1.so OBJDIR=build/v1 EXTRA=-DWT_BRANCH_WAVES=8)
EXTRA   ?=
FLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $(EXTRA)

all:
	$(MAKE) -j8 $(LIB)
	$(MAKE) oracle

$(OBJDIR)/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -x hip -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lhiprtc

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIB) build; $(MAKE) -C oracle clean

.PHONY: all oracle clean
