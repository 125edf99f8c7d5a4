# The micro-benchmark probes under scripts/*.hip (what an instruction, a barrier, an exchange between workgroups costs on
# gfx950): built into build/probes/ at the repository root -- not beside the sources, not into the package -- and run on
# the GPU box with `gpurun`.   make -f scripts/probes.mk            (from the repository root)
HIPCC ?= /opt/rocm/bin/hipcc
OUT   := build/probes
SRCS  := $(wildcard scripts/*_probe.hip)
BINS  := $(SRCS:scripts/%.hip=$(OUT)/%)
all: $(BINS)
$(OUT)/%: scripts/%.hip
	@mkdir -p $(OUT)
	$(HIPCC) -O3 --offload-arch=gfx950 -o $@ $<
clean:
	rm -rf $(OUT)
