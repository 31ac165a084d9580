# Convenience targets; the driver's contract is __graft_entry__.build() / smoke() and bench.py.
PY ?= python

.PHONY: build test test-gpu bench smoke clean

build:            ## libss_verify.so (hipcc --offload-arch=gfx950, cross-compiles without a GPU) + the oracle (tests only)
	$(PY) -c "import __graft_entry__ as g; g.build()"

test: build       ## CPU suite: oracle KATs, formats, native text readers (incl. ASan/UBSan fuzz), host logic, C ABI symbols
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build   ## parity suite on an MI355X
	$(PY) -m pytest tests -q -m gpu

smoke: build
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench: build      ## one JSON line: proofs/s on the 2^20 stwo configuration, roofline, e2e, cpu_baseline
	$(PY) bench.py

clean:
	$(MAKE) -C stark-symphony_amd/csrc clean
	$(MAKE) -C oracle clean
