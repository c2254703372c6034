# Build libprcg.so (gfx950 only) and the oracle's C helpers.  `make` or `python -c
# "import __graft_entry__ as g; g.build()"`.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := new_cg_variants_amd/csrc
OUT := new_cg_variants_amd/libprcg.so
# -ffp-contract=off: the reference's arithmetic is "multiply, round, add, round"
# (NumPy ufuncs, scipy csr_matvec); an FMA would change the bits.
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Iinclude
# TEST INFRASTRUCTURE: a stand-in for librccl.so that connects ranks living in threads of one process on one GPU
TRANSPORT := tests/transport/libthreads_ccl.so
# ... and one that connects ranks living in separate PROCESSES that share one GPU (the driver's launch shape, rehearsed on one GPU)
TRANSPORT_P := tests/transport/libprocs_ccl.so
OBJS := $(CSRC)/prcg_kernels.o $(CSRC)/prcg_win.o $(CSRC)/prcg_sell.o $(CSRC)/prcg_medium.o $(CSRC)/prcg_engine.o $(CSRC)/prcg_plan.o $(CSRC)/prcg_rccl.o

all: $(OUT) $(TRANSPORT) $(TRANSPORT_P)

$(CSRC)/prcg_kernels.o: $(CSRC)/prcg_kernels.hip $(CSRC)/prcg_kernels.h $(CSRC)/prcg_device.hpp
	$(HIPCC) --offload-arch=$(ARCH) $(CXXFLAGS) -c $< -o $@

$(CSRC)/prcg_win.o: $(CSRC)/prcg_win.hip $(CSRC)/prcg_kernels.h $(CSRC)/prcg_device.hpp
	$(HIPCC) --offload-arch=$(ARCH) $(CXXFLAGS) -c $< -o $@

$(CSRC)/prcg_sell.o: $(CSRC)/prcg_sell.hip $(CSRC)/prcg_kernels.h $(CSRC)/prcg_device.hpp
	$(HIPCC) --offload-arch=$(ARCH) $(CXXFLAGS) -c $< -o $@

$(CSRC)/prcg_medium.o: $(CSRC)/prcg_medium.hip $(CSRC)/prcg_kernels.h $(CSRC)/prcg_device.hpp
	$(HIPCC) --offload-arch=$(ARCH) $(CXXFLAGS) -c $< -o $@

$(CSRC)/prcg_engine.o: $(CSRC)/prcg_engine.cpp $(CSRC)/prcg_kernels.h $(CSRC)/prcg_plan.h $(CSRC)/prcg_rccl.h include/prcg.h include/prcg_test.h
	$(HIPCC) --offload-arch=$(ARCH) $(CXXFLAGS) -c $< -o $@

$(CSRC)/prcg_plan.o: $(CSRC)/prcg_plan.cpp $(CSRC)/prcg_plan.h
	$(HIPCC) --offload-arch=$(ARCH) $(CXXFLAGS) -c $< -o $@

$(CSRC)/prcg_rccl.o: $(CSRC)/prcg_rccl.cpp $(CSRC)/prcg_rccl.h
	$(HIPCC) --offload-arch=$(ARCH) $(CXXFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

$(TRANSPORT): tests/transport/threads_ccl.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -fPIC -shared $< -o $@

$(TRANSPORT_P): tests/transport/procs_ccl.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -fPIC -shared $< -o $@ -lrt

clean:
	rm -f $(OBJS) $(OUT) $(TRANSPORT) $(TRANSPORT_P)

.PHONY: all clean
