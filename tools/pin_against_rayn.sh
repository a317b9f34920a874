#!/bin/sh
# Pin the CPU oracle (oracle/rayn_oracle.cpp) against REAL rayn - one command, for whoever has a Rust toolchain (this repo's build
# environment has none: SURVEY.md F4, so parity of the oracle with rayn is "unpinned").  No GPU and no
# librayn_hip.so are needed: only bindings/rayn_dump.patch (the RAYN_DUMP hooks, nothing else) is applied to a COPY of the checkout.
#
#   tools/pin_against_rayn.sh <rayn checkout> [W H SAMPLES BOUNCES]        defaults: rayn's shipped 1280 720 2 3 (src/setup.rs:16-30)
#   TILE=<n> (default 1) selects the tile whose per-depth packets are traced; CARGO_FLAGS (default --release); KEEP=1 keeps the work dir
#   MOCK_RAYN=same|tables|fma|<variant>   dry run WITHOUT a Rust toolchain: every step below runs on the real checkout (copy, patch, the three
#       constant edits - each verified) EXCEPT `cargo run`, whose dump is written by the oracle playing rayn under that reading
#       (tools/rayn_dump.py mockrayn).  The outcome is known in advance - same: PINNED at stage 1; tables: PINNED at stage 2; fma / a variant:
#       stage 3 reports that reading IDENTICAL and the script exits 1 naming it - so the day a toolchain exists the script cannot fail for
#       script reasons (tests/test_evidence_tools.py runs all three outcomes at a small size; profiles/r05_pin_mock_shipped.txt at rayn's own).
#
# Steps: copy the checkout, apply the patch, set RESOLUTION / SAMPLES / MAX_INDIRECT_BOUNCES, `RAYN_DUMP=<dir>,<tile> cargo run`,
# dump the same frame from the oracle, compare array by array.  If the FILM differs the script goes on by itself:
#   stage 2  the oracle re-renders with rayn's OWN tables (--tables-from): differences left are not A6 / A7 (quasi-rd, SmallRng);
#   stage 3  every alternative reading (fused mul_add = A1, oracle/Makefile `variants` = A2..A5) is rendered with rayn's tables and
#            compared: the variant that comes out IDENTICAL (or closest) names the assumption that was read wrong.
# oracle/SENSITIVITY.md lists how far each reading moves a frame, i.e. which to suspect first.
set -e
RAYN=${1:?usage: pin_against_rayn.sh <rayn checkout> [W H SAMPLES BOUNCES]}
W=${2:-1280}; H=${3:-720}; S=${4:-2}; B=${5:-3}; TILE=${TILE:-1}
REPO=$(cd "$(dirname "$0")/.." && pwd)
[ -n "$MOCK_RAYN" ] || command -v cargo >/dev/null || { echo "cargo not found: this script needs a Rust toolchain (MOCK_RAYN=<reading> runs every other step without one)" >&2; exit 3; }
WORK=$(mktemp -d)
[ -n "$KEEP" ] || trap 'rm -rf "$WORK"' EXIT
cp -r "$RAYN" "$WORK/rayn"
cd "$WORK/rayn"
git apply -p1 "$REPO/bindings/rayn_dump.patch" 2>/dev/null || patch -p1 < "$REPO/bindings/rayn_dump.patch"
grep -q "^mod dump;" src/main.rs && [ -f src/dump.rs ] || { echo "bindings/rayn_dump.patch did not apply to $RAYN" >&2; exit 4; }
sed -i "s/pub const RESOLUTION: Extent2u = Extent2u::new([0-9]*, [0-9]*);/pub const RESOLUTION: Extent2u = Extent2u::new($W, $H);/; \
        s/pub const SAMPLES: usize = [0-9]*;/pub const SAMPLES: usize = $S;/; \
        s/pub const MAX_INDIRECT_BOUNCES: usize = [0-9]*;/pub const MAX_INDIRECT_BOUNCES: usize = $B;/" src/setup.rs
# the three edits must have taken (another rayn revision may spell the constants differently: fail here, not with a size mismatch later)
grep -q "pub const RESOLUTION: Extent2u = Extent2u::new($W, $H);" src/setup.rs && grep -q "pub const SAMPLES: usize = $S;" src/setup.rs &&
  grep -q "pub const MAX_INDIRECT_BOUNCES: usize = $B;" src/setup.rs || { echo "could not set RESOLUTION / SAMPLES / MAX_INDIRECT_BOUNCES in src/setup.rs" >&2; exit 4; }
mkdir -p renders
PY=${PYTHON:-python3}
DUMP="$PY $REPO/tools/rayn_dump.py"
ARGS="--scene ship --w $W --h $H --samples $S --bounces $B --tile $TILE"
if [ -n "$MOCK_RAYN" ]; then
  echo "== MOCK: no cargo run; the oracle writes rayn's dump as '$MOCK_RAYN'"
  $DUMP mockrayn "$WORK/dump_rayn" --as "$MOCK_RAYN" $ARGS
else
  RAYN_DUMP="$WORK/dump_rayn,$TILE" cargo run ${CARGO_FLAGS:---release}
fi
echo "== stage 1: oracle (default readings, its own tables) vs rayn"
$DUMP dump "$WORK/dump_oracle" $ARGS
if $DUMP compare "$WORK/dump_rayn" "$WORK/dump_oracle"; then echo "PINNED: the oracle reproduces rayn bit for bit on this frame"; exit 0; fi
echo "== stage 2: the oracle with rayn's own tables (A6 / A7 out of the picture)"
$DUMP dump "$WORK/dump_oracle_t" $ARGS --tables-from "$WORK/dump_rayn"
if $DUMP compare "$WORK/dump_rayn" "$WORK/dump_oracle_t"; then echo "PINNED up to the host tables: pass rayn's tables to rayn_hip_render_frame (they are plain inputs)"; exit 0; fi
echo "== stage 3: alternative readings, each with rayn's tables"
# order = oracle/SENSITIVITY.md's, shipped frame: the readings that move the most pixels beyond 1e-4 first - the FORM of normals_fast (central differences: 48 %), the
# mul_add policy (38 %), dot / normalized (32 / 31 %), the tetrahedron's summation order (9 %), libm (0.8 %), lerp (0.3 %); the max / min readings move nothing
FOUND=""
$DUMP dump "$WORK/dump_normals_central" $ARGS --tables-from "$WORK/dump_rayn" --variant normals_central
echo "-- normals_central"; if $DUMP compare "$WORK/dump_rayn" "$WORK/dump_normals_central" > "$WORK/cmp.txt"; then FOUND="$FOUND normals_central"; fi; tail -6 "$WORK/cmp.txt"
$DUMP dump "$WORK/dump_fma" $ARGS --tables-from "$WORK/dump_rayn" --fma 1
echo "-- fused mul_add (A1)"; if $DUMP compare "$WORK/dump_rayn" "$WORK/dump_fma" > "$WORK/cmp.txt"; then FOUND="$FOUND fma"; fi; tail -6 "$WORK/cmp.txt"
for V in dot_plain normalize_div normals_order libm lerp_alt minmax_swapped minmax_ieee; do
  $DUMP dump "$WORK/dump_$V" $ARGS --tables-from "$WORK/dump_rayn" --variant $V
  echo "-- $V"; if $DUMP compare "$WORK/dump_rayn" "$WORK/dump_$V" > "$WORK/cmp.txt"; then FOUND="$FOUND $V"; fi; tail -6 "$WORK/cmp.txt"
done
if [ -n "$FOUND" ]; then echo "NOT PINNED under the default readings, but IDENTICAL under:$FOUND - that assumption of oracle/rayn_oracle.cpp is read the other way by rayn"; exit 1; fi
echo "NOT PINNED: no single alternative reading reproduces rayn; see the per-array reports above (work dir: $WORK; re-run with KEEP=1 to keep it)"
exit 1
