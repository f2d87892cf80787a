# Build of the MI355X TetRex query engine.  gfx950 only; hipcc cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
CSRC    := tetrex_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
# `make EXPERIMENTS=1` (after `make clean`): timing experiments that deliberately compute WRONG masks (TXQ_HIBF_STORE bits 4/5,
# tools/ab_hibf*.sh) are compiled in.  The product build does not contain them.
ifdef EXPERIMENTS
HIPFLAGS += -DTXQ_EXPERIMENTS $(EXPERIMENT_FLAGS)
endif
HIP_SRCS := $(CSRC)/txq_api.hip $(CSRC)/txq_probe.hip $(CSRC)/txq_hibf.hip $(CSRC)/txq_exec.hip
HIP_OBJS := $(HIP_SRCS:.hip=.o)
HIP_HDRS := $(wildcard $(CSRC)/*.hpp) include/txq.h include/txq_program.h

HOST_DIR  := $(CSRC)/host
HOST_SRCS := $(HOST_DIR)/encoder.cpp $(HOST_DIR)/regex_front.cpp $(HOST_DIR)/kgraph.cpp $(HOST_DIR)/compiler.cpp $(HOST_DIR)/staged.cpp $(HOST_DIR)/index_file.cpp $(HOST_DIR)/matcher.cpp $(HOST_DIR)/host_capi.cpp
CLI_SRCS  := $(HOST_DIR)/main.cpp $(HOST_DIR)/device_index.cpp $(HOST_DIR)/verify.cpp $(HOST_DIR)/fasta.cpp
# (the verification matcher, matcher.cpp, is part of the host library so that tests can reach it through txh_regex_find_all)
HOST_HDRS := $(wildcard $(HOST_DIR)/*.hpp) include/txh.h include/txq_program.h
HOSTFLAGS := -O2 -g -std=c++20 -fPIC -Wall -Wextra -pthread

all: tetrex_amd/libtxq.so tetrex_amd/libtetrex_host.so tetrex_amd/libtetrex_query.so bin/tetrex oracle/liboracle.so

# whole-query execution on a GPU-resident index (host front-end + libtxq session), for bindings
QUERY_SRCS := $(HOST_DIR)/query_capi.cpp $(HOST_DIR)/device_index.cpp $(HOST_DIR)/fasta.cpp
tetrex_amd/libtetrex_query.so: $(QUERY_SRCS) $(HOST_SRCS) $(HOST_HDRS) tetrex_amd/libtxq.so tetrex_amd/libtetrex_host.so
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(QUERY_SRCS) -Ltetrex_amd -ltetrex_host -ltxq -lz \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,--enable-new-dtags

# The `tetrex` CLI: C++ host + libtxq.so (GPU) + zlib; the HIP runtime comes in through libtxq.so.
bin/tetrex: $(CLI_SRCS) $(HOST_SRCS) $(HOST_HDRS) tetrex_amd/libtxq.so tetrex_amd/libtetrex_host.so
	mkdir -p bin
	$(CXX) $(HOSTFLAGS) -fopenmp -o $@ $(CLI_SRCS) -Ltetrex_amd -ltetrex_host -ltxq -lz \
	    -Wl,-rpath,'$$ORIGIN/../tetrex_amd' -Wl,-rpath,/opt/rocm/lib -Wl,--enable-new-dtags

tetrex_amd/libtetrex_host.so: $(HOST_SRCS) $(HOST_HDRS)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(HOST_SRCS)

$(CSRC)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# The HIP runtime is linked by its UNVERSIONED name (libamdhip64.so) so that a process which
# already carries a HIP runtime under that name (PyTorch-ROCm bundles one) shares it instead
# of loading a second copy; standalone use resolves it through the rpath to /opt/rocm/lib.
tetrex_amd/libtxq.so: $(HIP_OBJS) $(CSRC)/hiprt_stub/libamdhip64.so
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -no-hip-rt -o $@ $(HIP_OBJS) \
	    -L$(CSRC)/hiprt_stub -lamdhip64 -Wl,-rpath,/opt/rocm/lib -Wl,--enable-new-dtags

# A link-time-only stand-in that carries no SONAME, so DT_NEEDED becomes "libamdhip64.so".
# It is never loaded: at run time the real runtime is found by name.
$(CSRC)/hiprt_stub/libamdhip64.so:
	mkdir -p $(CSRC)/hiprt_stub
	echo "" | $(CXX) -shared -fPIC -x c - -o $@

oracle/liboracle.so: $(wildcard oracle/*.hpp) oracle/txo_capi.cpp
	$(MAKE) -C oracle liboracle.so

clean:
	rm -f $(HIP_OBJS) tetrex_amd/libtxq.so tetrex_amd/libtetrex_host.so tetrex_amd/libtetrex_query.so bin/tetrex
	rm -rf $(CSRC)/hiprt_stub
	$(MAKE) -C oracle clean
.PHONY: all clean
