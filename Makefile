# Build of the MI355X ISSL scorer: libissl_hip.so (C ABI, include/issl_hip.h) and the two
# drop-in executables.  hipcc cross-compiles for gfx950 without a GPU present.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC     = crackling_amd/csrc
# -ffp-contract=off: the reference is built without FMA (Makefile:5, x86-64 baseline); the
# MIT/CFD doubles must round the same way on host and device.
CXXFLAGS = -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result
HIPFLAGS = $(CXXFLAGS) --offload-arch=$(ARCH)
LIB      = crackling_amd/libissl_hip.so

all: $(LIB) bin/isslScoreOfftargets bin/isslCreateIndex bin/extractOfftargets

$(LIB): $(CSRC)/issl_kernels.hip $(CSRC)/issl_extract.hip $(CSRC)/issl_build.hip $(CSRC)/issl_capi.cpp $(CSRC)/issl_host.cpp $(CSRC)/issl_text.cpp \
        $(CSRC)/issl_node.cpp $(CSRC)/issl_host.hpp $(CSRC)/issl_device.hpp $(CSRC)/issl_radix.hpp $(CSRC)/cfd_tables.inc include/issl_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/issl_kernels.hip $(CSRC)/issl_extract.hip $(CSRC)/issl_build.hip $(CSRC)/issl_capi.cpp \
	    $(CSRC)/issl_host.cpp $(CSRC)/issl_text.cpp $(CSRC)/issl_node.cpp -lpthread -ldl

# host-only executable: libissl_hip.so is loaded with dlopen when the process has to score by itself, not when a resident
# server answers (cli_score.cpp)
bin/isslScoreOfftargets: $(CSRC)/cli_score.cpp include/issl_hip.h $(LIB)
	@mkdir -p bin
	g++ $(CXXFLAGS) -o $@ $< -lpthread -ldl

bin/extractOfftargets: $(CSRC)/cli_extract.cpp $(LIB)
	@mkdir -p bin
	$(HIPCC) $(CXXFLAGS) -o $@ $< -Lcrackling_amd -lissl_hip -Wl,-rpath,'$$ORIGIN/../crackling_amd'

# host-only executable: no HIP runtime behind it (start-up of libamdhip64 alone costs ~1.5 s)
bin/isslCreateIndex: $(CSRC)/cli_create.cpp $(CSRC)/issl_host.cpp $(CSRC)/issl_host.hpp
	@mkdir -p bin
	g++ $(CXXFLAGS) -o $@ $(CSRC)/cli_create.cpp $(CSRC)/issl_host.cpp -lpthread

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(LIB) bin
.PHONY: all oracle clean
