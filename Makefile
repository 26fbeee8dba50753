# Top-level conveniences.  `make` builds what __graft_entry__.build() builds; `make san` is the host-side sanitizer
# build + run (CPU container only: SURVEY.md 5 "sanitizers"; GPU sanitizers are not available on the pool).
PKG := lattice-boltzmann-method_amd
SAN_RT := $(shell /opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)

all:
	$(MAKE) -C $(PKG)/csrc -j8
	$(MAKE) -C $(PKG)/drivers -j8
	$(MAKE) -C oracle

# ASan + UBSan over everything of the C ABI and the test infrastructure that a CPU can reach: the oracle (every function the
# golden vectors exercise), the C ABI's host side through ctypes (validation of every entry point's arguments, the
# slab-height planner, link tables, tuning table), the TOML reader and the params:: mirror (params_dump).
san:
	$(MAKE) -C $(PKG)/csrc SAN=1 -j8
	$(MAKE) -C oracle SAN=1
	$(MAKE) -C $(PKG)/drivers bin_san/params_dump
	mkdir -p profiles
	LD_PRELOAD=$(SAN_RT) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	LBM_HIP_LIB=$(CURDIR)/$(PKG)/lib_san/liblbm_hip.so LBM_ORACLE_LIB=$(CURDIR)/oracle/_build_san/liblbm_oracle.so \
	LBM_PARAMS_DUMP=$(CURDIR)/$(PKG)/drivers/bin_san/params_dump OMP_NUM_THREADS=4 \
	python -m pytest tests/test_oracle_golden.py tests/test_oracle_crosscheck.py tests/test_host_cpp.py tests/test_slab_plan.py tests/test_abi.py \
	  -q -m "not gpu" -p no:cacheprovider 2>&1 | tee profiles/r04_san_cpu.log
.PHONY: all san
