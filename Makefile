# Build libegnn_amd.so (gfx950 only).  `python -c "import __graft_entry__ as g; g.build()"` runs this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := diffusion_model_amd/csrc
OUT   := diffusion_model_amd/libegnn_amd.so
SRCS  := $(CSRC)/egnn_forward.hip $(CSRC)/edge_bf16_v3.hip $(CSRC)/edge_bf16_v4.hip $(CSRC)/edge_x_m16.hip $(CSRC)/edge_bf16x3.hip $(CSRC)/edge_bwd_dgrad.hip $(CSRC)/edge_bwd_heads.hip $(CSRC)/gemm_tn.hip $(CSRC)/gemm_rows.hip $(CSRC)/sampler.hip $(CSRC)/graph_stats.hip $(CSRC)/aux_mlp.hip $(CSRC)/node_bf16.hip $(CSRC)/backward.hip
HDRS  := $(CSRC)/common.h $(CSRC)/kernels.h $(CSRC)/edge_tile.h include/egnn_amd.h
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize

OBJS := $(SRCS:.hip=.o)

all: $(OUT)

%.o: %.hip $(HDRS)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

clean:
	rm -f $(OBJS) $(OUT)

.PHONY: all clean
