# Build libegnn_amd.so (gfx950 only).  `python -c "import __graft_entry__ as g; g.build()"` runs this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := diffusion_model_amd/csrc
OUT   := diffusion_model_amd/libegnn_amd.so
SRCS  := $(CSRC)/egnn_forward.hip $(CSRC)/edge_bf16_v3.hip $(CSRC)/edge_bf16_v4.hip $(CSRC)/edge_x_m16.hip $(CSRC)/edge_small.hip $(CSRC)/edge_bf16x3.hip $(CSRC)/edge_f16c8.hip $(CSRC)/edge_f16c8w.hip $(CSRC)/edge_bwd_dgrad.hip $(CSRC)/edge_bwd_dgrad_graph.hip $(CSRC)/edge_bwd_heads.hip $(CSRC)/edge_bwd_first.hip $(CSRC)/gemm_tn.hip $(CSRC)/gemm_rows.hip $(CSRC)/sampler.hip $(CSRC)/graph_stats.hip $(CSRC)/aux_mlp.hip $(CSRC)/node_bf16.hip $(CSRC)/backward.hip
HDRS  := $(CSRC)/edge_f16c8_mphase2.inc $(CSRC)/edge_f16c8_mphase4.inc $(CSRC)/edge_f16c8w_mphase1.inc $(CSRC)/edge_f16c8w_mphase2.inc $(CSRC)/edge_f16c8w_mphasek.inc $(CSRC)/common.h $(CSRC)/kernels.h $(CSRC)/edge_tile.h $(CSRC)/host_logic.h $(CSRC)/diag.h $(CSRC)/bwd_graph.h include/egnn_amd.h
# -fvisibility=hidden: the library exports exactly the functions include/egnn_amd.h declares (the header wraps its
# declarations in a visibility push(default)); tests/test_cabi_and_host.py compares the two sets
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden
# host_logic.cpp: the HIP-free part of the host side (validation, schedule builder, plans); plain C++ for both builds
HOSTSRC := $(CSRC)/host_logic.cpp
HOSTOBJ := $(CSRC)/host_logic.o
ASAN_OUT := build/libegnn_host_asan.so

OBJS := $(SRCS:.hip=.o)

all: $(OUT)

%.o: %.hip $(HDRS)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(HOSTOBJ): $(HOSTSRC) $(CSRC)/host_logic.h include/egnn_amd.h
	$(HIPCC) -O2 -std=c++17 -fPIC -Wall -fvisibility=hidden -c $< -o $@

$(OUT): $(OBJS) $(HOSTOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) $(HOSTOBJ) -o $@

# CPU-only sanitizer build of the host logic (no GPU, no HIP): what tests/test_host_asan.py loads in the build container
asan: $(ASAN_OUT)
$(ASAN_OUT): $(HOSTSRC) $(CSRC)/host_logic.h include/egnn_amd.h
	mkdir -p build
	g++ -O1 -g -std=c++17 -fPIC -shared -Wall -Wextra -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -DEGNN_HOST_TEST_API $(HOSTSRC) -o $@

clean:
	rm -f $(OBJS) $(HOSTOBJ) $(OUT) $(ASAN_OUT)

.PHONY: all clean asan
