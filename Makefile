# Native pieces (gfx950 only).  `python -c "import __graft_entry__ as g; g.build()"` does the same from Python.
HIPCC ?= hipcc
HIPFLAGS = -O3 -ffp-contract=off --offload-arch=gfx950 -std=c++17 -fPIC -shared

all: lib oracle ubench

lib: bayesflow_nddms_amd/libnddm_hip.so
bayesflow_nddms_amd/libnddm_hip.so: bayesflow_nddms_amd/csrc/nddm_kernels.hip bayesflow_nddms_amd/csrc/nddm_rng.h include/nddm.h
	$(HIPCC) $(HIPFLAGS) -o $@ $<

# test infrastructure only (CPU oracle); never linked into the product
oracle: oracle/liboracle.so
oracle/liboracle.so: oracle/ddm_oracle.c
	gcc -O2 -ffp-contract=off -mfma -fno-math-errno -fopenmp -fPIC -shared -o $@ $< -lm

ubench: tools/ubench_valu
tools/ubench_valu: tools/ubench_valu.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -o $@ $<

test-cpu:
	python -m pytest tests -x -q -m "not gpu"
test-gpu:
	python -m pytest tests -x -q -m gpu

clean:
	rm -f bayesflow_nddms_amd/libnddm_hip.so oracle/liboracle.so tools/ubench_valu
.PHONY: all lib oracle ubench test-cpu test-gpu clean
