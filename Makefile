# Builds everything in-tree (the built .so files travel to the GPU box with the source tree).
#   make -j8        product library + example-scene library + oracle + test helpers
#   make product    raytracer_amd/_lib/librt_amd.so (HIP, gfx950) and librt_scenes.so
# Experimental builds of the product library (scripts/exp_build.sh):
#   make -j8 kernels LIBDIR=scratch/exp/NAME OBJDIR=scratch/exp/NAME/obj EXTRA_HIPFLAGS="-DRT_EXP=4"
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
LIBDIR   ?= raytracer_amd/_lib
OBJDIR   ?= build/obj
# -memssa-check-limit: the scalar (s_load) fetches of the scene records need the compiler's proof that no store of the
# kernel can clobber them, and that proof walks MemorySSA with a budget of 100 steps by default; one more LDS store in the
# path loop and every record of the union walks came through vector loads (cover_image 1.33 -> 1.86 ms).  DESIGN.md §7.2.
HIPCFLAGS := --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical \
             -mllvm -memssa-check-limit=2000
CXXFLAGS := -std=c++17 -O2 -fPIC -pthread -Wall -Wextra -Wno-unused-parameter -ffp-contract=off
CSRC     := raytracer_amd/csrc
KHEADERS := $(CSRC)/rt_kernel.h $(CSRC)/rt_math.h $(CSRC)/rt_exact.h include/rt_amd.h
# render_kernel's instantiations, one translation unit per group (rt_kernels.hip): they compile in parallel
KGROUPS  := 0 1 3 4 5 7
KOBJS    := $(KGROUPS:%=$(OBJDIR)/rt_kernels_%.o)

all: product oracle testhelpers
product: $(LIBDIR)/librt_amd.so $(LIBDIR)/librt_scenes.so
kernels: $(LIBDIR)/librt_amd.so

$(OBJDIR)/rt_kernels_%.o: $(CSRC)/rt_kernels.hip $(KHEADERS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPCFLAGS) $(EXTRA_HIPFLAGS) -DRT_KGROUP=$* -c $(CSRC)/rt_kernels.hip -o $@

$(OBJDIR)/rt_amd.o: $(CSRC)/rt_amd.hip $(CSRC)/rt_aux_kernels.h $(CSRC)/rt_pack.h $(KHEADERS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPCFLAGS) $(EXTRA_HIPFLAGS) -c $(CSRC)/rt_amd.hip -o $@

$(LIBDIR)/librt_amd.so: $(OBJDIR)/rt_amd.o $(KOBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -pthread -Wl,-Bsymbolic $^ -o $@

$(LIBDIR)/librt_scenes.so: scenes/scenes_capi.cpp scenes/examples.inc include/raytracer_amd/raytracer.hpp include/raytracer_amd/png.hpp include/rt_amd.h $(LIBDIR)/librt_amd.so
	$(CXX) $(CXXFLAGS) -shared scenes/scenes_capi.cpp -o $@ -L$(LIBDIR) -lrt_amd -lz -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

testhelpers: tests/cpp/_build/libhostsim.so
tests/cpp/_build/libhostsim.so: tests/cpp/hostsim.cpp raytracer_amd/csrc/rt_math.h raytracer_amd/csrc/rt_exact.h raytracer_amd/csrc/rt_pack.h include/rt_amd.h
	@mkdir -p tests/cpp/_build
	# -Bsymbolic: the simulator must run ITS OWN copies of the inline functions of rt_math.h / rt_pack.h,
	# not the ones exported by the product library that the same test process has loaded (RTLD_GLOBAL)
	$(CXX) $(CXXFLAGS) -shared -Wl,-Bsymbolic tests/cpp/hostsim.cpp -o $@

clean:
	rm -rf $(LIBDIR) $(OBJDIR) tests/cpp/_build
	$(MAKE) -C oracle clean
.PHONY: all product kernels oracle testhelpers clean
