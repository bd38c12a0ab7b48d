# Build libclipk.so (gfx950 HIP kernels + C ABI) in-tree, and the oracle's C pieces.
#   make            -> clip_dplm_amd/lib/libclipk.so
#   make probes     -> tools/probes/probe_layouts (hardware layout probes, run on the GPU box)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wno-unused-value -Iinclude
SRC := $(wildcard clip_dplm_amd/csrc/*.hip)
OBJ := $(patsubst clip_dplm_amd/csrc/%.hip,build/%.o,$(SRC))
LIB := clip_dplm_amd/lib/libclipk.so

all: $(LIB)

build/%.o: clip_dplm_amd/csrc/%.hip clip_dplm_amd/csrc/common.h clip_dplm_amd/csrc/gemm_epilogue.h include/clipk.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p clip_dplm_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

probes: tools/probes/probe_layouts tools/probes/probe_gather tools/probes/probe_store_shapes tools/probes/probe_store_vs_dma tools/probes/probe_mfma_valu
tools/probes/probe_mfma_valu: tools/probes/probe_mfma_valu.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value -o $@ $<
tools/probes/probe_store_vs_dma: tools/probes/probe_store_vs_dma.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value -o $@ $<
# experiment build of the weight-gradient kernel with barrier cycle stamps (tools/exp_wgrad_trace.py)
tools/probes/libwgrad_trace.so: clip_dplm_amd/csrc/gemm_wgrad_v3.hip clip_dplm_amd/csrc/core.hip clip_dplm_amd/csrc/common.h
	$(HIPCC) $(HIPFLAGS) -DCLIPK_WGRAD_TRACE -shared -o $@ clip_dplm_amd/csrc/gemm_wgrad_v3.hip clip_dplm_amd/csrc/core.hip
# ... and its timing ablations (results garbage): make tools/probes/libwgrad_trace_abl3.so
tools/probes/libwgrad_trace_abl%.so: clip_dplm_amd/csrc/gemm_wgrad_v3.hip clip_dplm_amd/csrc/core.hip clip_dplm_amd/csrc/common.h
	$(HIPCC) $(HIPFLAGS) -DCLIPK_WGRAD_TRACE -DCLIPK_WGRAD_ABL=$* -shared -o $@ clip_dplm_amd/csrc/gemm_wgrad_v3.hip clip_dplm_amd/csrc/core.hip
# experiment builds of the 256 x 256 Linear kernel with in-kernel clock stamps (tools/exp_gemm_mfma_shape.py):
# libgemm_trace16.so = the product's 16x16x32 main loop, libgemm_trace32.so = the same loop issuing 32x32x16 MFMAs
# (timing only: results garbage).  Both run WITHOUT the epilogue (option gemm_abl = 1): the main loop alone.
tools/probes/libgemm_trace16.so: clip_dplm_amd/csrc/gemm_nt_v3.hip clip_dplm_amd/csrc/core.hip clip_dplm_amd/csrc/common.h clip_dplm_amd/csrc/gemm_epilogue.h
	$(HIPCC) $(HIPFLAGS) -DCLIPK_EXPERIMENTS -DCLIPK_GEMM_TRACE -shared -o $@ clip_dplm_amd/csrc/gemm_nt_v3.hip clip_dplm_amd/csrc/core.hip
tools/probes/libgemm_trace32.so: clip_dplm_amd/csrc/gemm_nt_v3.hip clip_dplm_amd/csrc/core.hip clip_dplm_amd/csrc/common.h clip_dplm_amd/csrc/gemm_epilogue.h
	$(HIPCC) $(HIPFLAGS) -DCLIPK_EXPERIMENTS -DCLIPK_GEMM_TRACE -DCLIPK_GEMM_MFMA32 -shared -o $@ clip_dplm_amd/csrc/gemm_nt_v3.hip clip_dplm_amd/csrc/core.hip
# experiment build of the attention kernels with phase stamps in the whole-head backward (tools/exp_attn_trace.py)
tools/probes/libattn_trace.so: clip_dplm_amd/csrc/attention.hip clip_dplm_amd/csrc/core.hip clip_dplm_amd/csrc/common.h
	$(HIPCC) $(HIPFLAGS) -DCLIPK_ATTN_TRACE -shared -o $@ clip_dplm_amd/csrc/attention.hip clip_dplm_amd/csrc/core.hip
# timing-only build: q / k / v / dO of the whole-head kernels addressed head-major (tools/exp_attn_headmajor.py)
tools/probes/libattn_hm.so: clip_dplm_amd/csrc/attention.hip clip_dplm_amd/csrc/core.hip clip_dplm_amd/csrc/common.h
	$(HIPCC) $(HIPFLAGS) -DCLIPK_ATTN_HM_PROBE -shared -o $@ clip_dplm_amd/csrc/attention.hip clip_dplm_amd/csrc/core.hip
# experiment build of the whole library (timing ablations behind option gemm_abl; results garbage):
#   BENCH_LIB=tools/probes/libclipk_exp.so BENCH_ABL="1 4" python3 tools/bench_kernels.py gemm
tools/probes/libclipk_exp.so: $(SRC) $(wildcard clip_dplm_amd/csrc/*.h) include/clipk.h
	$(HIPCC) $(HIPFLAGS) -DCLIPK_EXPERIMENTS -shared -o $@ $(SRC)
tools/probes/probe_layouts: tools/probes/probe_layouts.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -Wno-unused-value -o $@ $<
tools/probes/probe_gather: tools/probes/probe_gather.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value -o $@ $<
tools/probes/probe_store_shapes: tools/probes/probe_store_shapes.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value -o $@ $<

clean:
	rm -rf build $(LIB) tools/probes/probe_layouts tools/probes/probe_gather tools/probes/probe_store_shapes tools/probes/probe_store_vs_dma tools/probes/probe_mfma_valu

.PHONY: all clean probes
