# Native pieces (gfx950 only).  `python -c "import __graft_entry__ as g; g.build()"` does the same from Python.
HIPCC ?= hipcc
HIPFLAGS = -O3 -ffp-contract=off --offload-arch=gfx950 -std=c++17 -fPIC -shared

all: lib oracle ubench demo

lib: bayesflow_nddms_amd/libnddm_hip.so
bayesflow_nddms_amd/libnddm_hip.so: bayesflow_nddms_amd/csrc/nddm_kernels.hip bayesflow_nddms_amd/csrc/nddm_sim.h bayesflow_nddms_amd/csrc/nddm_prepass.h bayesflow_nddms_amd/csrc/nddm_ratcliff.h bayesflow_nddms_amd/csrc/nddm_rng.h include/nddm.h
	python -m bayesflow_nddms_amd.build    # (hipcc $(HIPFLAGS) + the content hash of the sources, -DNDDM_SOURCE_HASH)

# test infrastructure only (CPU oracle); never linked into the product
oracle: oracle/liboracle.so
oracle/liboracle.so: oracle/ddm_oracle.c
	gcc -O2 -ffp-contract=off -mfma -fno-math-errno -fopenmp -fPIC -shared -o $@ $< -lm

ubench: tools/ubench_valu tools/ubench_residency tools/ubench_bank
tools/ubench_valu: tools/ubench_valu.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -o $@ $<
tools/ubench_residency: tools/ubench_residency.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -o $@ $<
tools/ubench_bank: tools/ubench_bank.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -o $@ $<

# the boundary from plain C (examples/c_abi_demo.c): no Python, no PyTorch
demo: examples/c_abi_demo
examples/c_abi_demo: examples/c_abi_demo.c include/nddm.h bayesflow_nddms_amd/libnddm_hip.so
	gcc -O2 -std=c99 -D_POSIX_C_SOURCE=199309L -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude $< -o $@ \
	    -Lbayesflow_nddms_amd -lnddm_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../bayesflow_nddms_amd' -Wl,-rpath,/opt/rocm/lib

test-cpu:
	python -m pytest tests -x -q -m "not gpu"
test-gpu:
	python -m pytest tests -x -q -m gpu

clean:
	rm -f bayesflow_nddms_amd/libnddm_hip.so oracle/liboracle.so tools/ubench_valu tools/ubench_residency tools/ubench_bank examples/c_abi_demo
.PHONY: all lib oracle ubench demo test-cpu test-gpu clean
