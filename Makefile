# Builds everything that ships: the host scene library (g++), the HIP render-loop library
# (hipcc, gfx950) and the command-line front-end. `make oracle` builds the test-only CPU restatement.
ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
PKG     := $(ROOT)raytracer_amd
HOSTDIR := $(PKG)/host
CSRC    := $(PKG)/csrc
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
# No fused multiply-add anywhere a hit decision is made: the reference's x86-64 build has none.
HOSTFLAGS := -std=c++17 -O2 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -I$(ROOT)include -I$(HOSTDIR)
HIPFLAGS  := -std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $(HIPEXTRA) -I$(ROOT)include -I$(CSRC) \
             -Wall -Wno-unused-parameter

HOST_SRCS := $(HOSTDIR)/rth_scene.cpp $(HOSTDIR)/rth_bvh.cpp $(HOSTDIR)/rth_parse.cpp $(HOSTDIR)/rth_png.cpp $(HOSTDIR)/rth_capi.cpp
HOST_HDRS := $(HOSTDIR)/rth_scene.h $(HOSTDIR)/rth_math.h $(HOSTDIR)/rth_threads.h $(ROOT)include/rt_amd.h $(ROOT)include/rt_host.h
HIP_SRCS  := $(CSRC)/rt_device.hip $(CSRC)/rt_bvh_build.hip
HIP_HDRS  := $(wildcard $(CSRC)/*.h) $(ROOT)include/rt_amd.h

.PHONY: all host hip cli oracle ref clean
all: host hip cli

host: $(PKG)/librt_host.so
hip: $(PKG)/libraytracer_amd.so
cli: $(PKG)/raytracer

$(PKG)/librt_host.so: $(HOST_SRCS) $(HOST_HDRS)
	$(CXX) $(HOSTFLAGS) -pthread -shared $(HOST_SRCS) -o $@ -lz

$(PKG)/libraytracer_amd.so: $(HIP_SRCS) $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -shared $(HIP_SRCS) -o $@

$(PKG)/raytracer: $(HOSTDIR)/rth_main.cpp $(PKG)/librt_host.so $(PKG)/libraytracer_amd.so
	$(CXX) $(HOSTFLAGS) $< -o $@ -L$(PKG) -lrt_host -lraytracer_amd -Wl,-rpath,'$$ORIGIN' -lpthread

oracle:
	$(MAKE) -C $(ROOT)oracle oracle

ref:
	$(MAKE) -C $(ROOT)oracle ref

clean:
	rm -f $(PKG)/librt_host.so $(PKG)/libraytracer_amd.so $(PKG)/raytracer
	$(MAKE) -C $(ROOT)oracle clean
