# Builds the product library (HIP, gfx950) and the test-infrastructure oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC = raytracer_amd/csrc
SRCS = $(CSRC)/scene_build.cpp $(CSRC)/bvh_build.hip $(CSRC)/intersect.hip $(CSRC)/walk_tree.hip $(wildcard $(CSRC)/render.hip)
HDRS = $(wildcard $(CSRC)/*.h) include/rtamd.h
# -ffp-contract=off: the intersector must round like the reference's scalar code (bit-exact triangle IDs)
# -fno-slp-vectorize: the SLP vectorizer packs the triangle test into v_pk_* ops glued by ~35 v_movs; measured 4 % slower
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function $(EXTRA)
OBJDIR ?= build
OBJS = $(patsubst $(CSRC)/%,$(OBJDIR)/%.o,$(SRCS))

all: product oracle host
# C++ host mirror + headless driver (links the C ABI only; no HIP code on this side)
host: raytracer_amd/host/rtamd_cli tests/host/rttest
raytracer_amd/host/rtamd_cli: raytracer_amd/host/main.cpp raytracer_amd/host/rtamd.hpp raytracer_amd/host/gem.hpp include/rtamd.h raytracer_amd/librtamd.so
	g++ -std=c++17 -O2 -ffp-contract=off -Wall -Wextra raytracer_amd/host/main.cpp -o $@ -Lraytracer_amd -lrtamd -Wl,-rpath,'$$ORIGIN/..'
LIB ?= raytracer_amd/librtamd.so
product: $(LIB)
# one object per translation unit (make -j builds them side by side), one link
$(OBJDIR)/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@
# an experiment build beside the product: make -j5 LIB=build_ab/w5.so OBJDIR=build_ab/w5 EXTRA=-DRT_TRAV_WAVES=5 product; tools/ab.sh runs the bench per library
oracle:
	$(MAKE) -C oracle
clean:
	rm -f raytracer_amd/librtamd.so oracle/liboracle.so raytracer_amd/host/rtamd_cli
.PHONY: all product oracle host clean
# the reference's RTTest known-answer case through the C++ host mirror (run by tests/test_host_cpp.py on a GPU)
tests/host/rttest: tests/host/rttest.cpp raytracer_amd/host/rtamd.hpp include/rtamd.h raytracer_amd/librtamd.so
	g++ -std=c++17 -O2 -ffp-contract=off -Wall -Wextra tests/host/rttest.cpp -o $@ -Lraytracer_amd -lrtamd -Wl,-rpath,'$$ORIGIN/../../raytracer_amd'
