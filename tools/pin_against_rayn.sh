#!/bin/sh
# Pin the CPU oracle (oracle/rayn_oracle.cpp) against REAL rayn - one command, for whoever has a Rust toolchain (this repo's build
# environment has none: SURVEY.md F4, so parity of the oracle with rayn is "unpinned").  No GPU and no
# librayn_hip.so are needed: only bindings/rayn_dump.patch (the RAYN_DUMP hooks, nothing else) is applied to a COPY of the checkout.
#
#   tools/pin_against_rayn.sh <rayn checkout> [W H SAMPLES BOUNCES]        defaults: rayn's shipped 1280 720 2 3 (src/setup.rs:16-30)
#   TILE=<n> (default 1) selects the tile whose per-depth packets are traced; CARGO_FLAGS (default --release); KEEP=1 keeps the work dir
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
command -v cargo >/dev/null || { echo "cargo not found: this script needs a Rust toolchain" >&2; exit 3; }
WORK=$(mktemp -d)
[ -n "$KEEP" ] || trap 'rm -rf "$WORK"' EXIT
cp -r "$RAYN" "$WORK/rayn"
cd "$WORK/rayn"
git apply -p1 "$REPO/bindings/rayn_dump.patch" 2>/dev/null || patch -p1 < "$REPO/bindings/rayn_dump.patch"
sed -i "s/pub const RESOLUTION: Extent2u = Extent2u::new([0-9]*, [0-9]*);/pub const RESOLUTION: Extent2u = Extent2u::new($W, $H);/; \
        s/pub const SAMPLES: usize = [0-9]*;/pub const SAMPLES: usize = $S;/; \
        s/pub const MAX_INDIRECT_BOUNCES: usize = [0-9]*;/pub const MAX_INDIRECT_BOUNCES: usize = $B;/" src/setup.rs
mkdir -p renders
RAYN_DUMP="$WORK/dump_rayn,$TILE" cargo run ${CARGO_FLAGS:---release}
PY=${PYTHON:-python3}
DUMP="$PY $REPO/tools/rayn_dump.py"
ARGS="--scene ship --w $W --h $H --samples $S --bounces $B --tile $TILE"
echo "== stage 1: oracle (default readings, its own tables) vs rayn"
$DUMP dump "$WORK/dump_oracle" $ARGS
if $DUMP compare "$WORK/dump_rayn" "$WORK/dump_oracle"; then echo "PINNED: the oracle reproduces rayn bit for bit on this frame"; exit 0; fi
echo "== stage 2: the oracle with rayn's own tables (A6 / A7 out of the picture)"
$DUMP dump "$WORK/dump_oracle_t" $ARGS --tables-from "$WORK/dump_rayn"
if $DUMP compare "$WORK/dump_rayn" "$WORK/dump_oracle_t"; then echo "PINNED up to the host tables: pass rayn's tables to rayn_hip_render_frame (they are plain inputs)"; exit 0; fi
echo "== stage 3: alternative readings, each with rayn's tables"
$DUMP dump "$WORK/dump_fma" $ARGS --tables-from "$WORK/dump_rayn" --fma 1
echo "-- fused mul_add (A1)"; $DUMP compare "$WORK/dump_rayn" "$WORK/dump_fma" | tail -6 || true
for V in minmax_swapped minmax_ieee libm normalize_div dot_plain normals_central normals_order lerp_alt; do
  $DUMP dump "$WORK/dump_$V" $ARGS --tables-from "$WORK/dump_rayn" --variant $V
  echo "-- $V"; $DUMP compare "$WORK/dump_rayn" "$WORK/dump_$V" | tail -6 || true
done
echo "NOT PINNED: see the per-array reports above (work dir: $WORK; re-run with KEEP=1 to keep it)"
exit 1
