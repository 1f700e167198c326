# Builds everything in-tree (the built .so files travel to the GPU box with the source tree).
#   make            product library + example-scene library + oracle + test helpers
#   make product    raytracer_amd/_lib/librt_amd.so (HIP, gfx950) and librt_scenes.so
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
LIBDIR   := raytracer_amd/_lib
HIPFLAGS := --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function
CXXFLAGS := -std=c++17 -O2 -fPIC -pthread -Wall -Wextra -Wno-unused-parameter -ffp-contract=off

all: product oracle testhelpers
product: $(LIBDIR)/librt_amd.so $(LIBDIR)/librt_scenes.so

$(LIBDIR)/librt_amd.so: raytracer_amd/csrc/rt_amd.hip raytracer_amd/csrc/rt_kernel.h raytracer_amd/csrc/rt_math.h raytracer_amd/csrc/rt_exact.h raytracer_amd/csrc/rt_pack.h include/rt_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_HIPFLAGS) -Wl,-Bsymbolic raytracer_amd/csrc/rt_amd.hip -o $@

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
	rm -rf $(LIBDIR) tests/cpp/_build
	$(MAKE) -C oracle clean
.PHONY: all product oracle testhelpers clean
