#!/bin/bash
# Optional CPU baseline on the REAL reference: builds dfyz/osm-renderer with cargo and times its own
# renderer on the host cores — only where that is possible.  It needs (1) cargo + rustc, (2) the reference checkout with a
# vendored crate registry (there is no network on the GPU boxes), (3) the reference's test fixture (tests/osm/*.osm, a
# large blob missing from this image).  Anything missing: prints SKIPPED and exits 0 — bench.py's cpu_baseline (the C++
# restatement in oracle/, stated as such) is the number that is always reported.
#
#   scripts/rust_baseline.sh [reference_dir]      default: /root/reference
REF=${1:-/root/reference}
skip() { echo "SKIPPED: $1"; exit 0; }
command -v cargo >/dev/null 2>&1 || skip "cargo not found (no Rust toolchain in this image)"
command -v rustc >/dev/null 2>&1 || skip "rustc not found"
[ -f "$REF/Cargo.toml" ] || skip "no reference checkout at $REF (the GPU boxes only receive /root/repo)"
[ -d "$REF/vendor" ] || [ -d "$HOME/.cargo/registry/cache" ] || skip "no vendored crate registry and no network"
ls "$REF"/tests/osm/*.osm >/dev/null 2>&1 || skip "tests/osm fixture missing (.MISSING_LARGE_BLOBS)"
set -e
cd "$REF"
CORES=$(nproc)
echo "building the reference with cargo (offline) on $CORES cores: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
cargo build --release --offline
# the reference's own rendering test renders the fixture's z14..z18 mosaics (tests/test_rendering.rs:147-176): time it
START=$(date +%s.%N)
cargo test --release --offline --test test_rendering -- --test-threads "$CORES"
END=$(date +%s.%N)
echo "reference test_rendering wall clock: $(echo "$END - $START" | bc) s on $CORES cores (Rust binary, not the C++ restatement)"
