#!/bin/bash
# One parametrised A/B runner for the GPU box (replaces the one-off tools/r05_run*.sh scripts of round 5).
#
#   tools/ab.sh <tag> [-m train|infer|skip] [-r REPS] [-s STEPS] [-t "pytest selection"] -- "<variant A>" "<variant B>" ...
#
# A variant is a string of environment assignments ("" = the defaults), e.g. "TTRAP_SKIP_FOLD=0" or "TTRAP_LIB=libttrap_x.so"
# (a library built with tools/build_variant.sh).  The variants run back to back on the same box, REPS rounds (default 2), each as a
# fresh `python bench.py` process: -m train (default) = `--timed-only` train step, -m infer = `--mode infer` (BASELINE configs[1]),
# -m skip = the full line's skip_connections_step (slower: runs every secondary leg).  -t: a pytest selection run first WITH EACH
# variant's environment (parity before speed).  Output: gpurun_out/<tag>.txt (one "variant | ms" line per run) -- copy into profiles/.
tag=$1; shift
mode=train; reps=2; steps=20; tests=""
while [ "$1" != "--" ] && [ $# -gt 0 ]; do
  case $1 in
    -m) mode=$2; shift 2;;
    -r) reps=$2; shift 2;;
    -s) steps=$2; shift 2;;
    -t) tests=$2; shift 2;;
    *) echo "unknown option $1" >&2; exit 2;;
  esac
done
shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/$tag.txt
echo "# tools/ab.sh $tag: mode $mode, $reps rounds, $(date -u +%FT%TZ), $(git rev-parse --short HEAD 2>/dev/null)" > $out
if [ -n "$tests" ]; then
  for v in "$@"; do
    echo "== tests [$v]: $tests" >> $out
    env $v python -m pytest $tests -q -m gpu -x --tb=short -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
    tail -2 gpurun_out/${tag}_tests.log >> $out
  done
fi
pick='import json,sys
d=json.loads(sys.stdin.readline())
m=sys.argv[1]
print("%.3f" % (d["skip_connections_step"]["ms_per_step"] if m=="skip" else d["ms_per_step"]))'
for i in $(seq 1 $reps); do
  for v in "$@"; do
    case $mode in
      train) args="--timed-only --no-cpu-baseline --steps $steps --warmup 5";;
      infer) args="--mode infer --steps $steps --warmup 3";;
      skip)  args="--no-cpu-baseline --steps 10 --warmup 5";;
    esac
    ms=$(env $v python bench.py $args 2>/dev/null | python -c "$pick" $mode)
    echo "[${v:-defaults}] | $ms ms" >> $out
  done
done
cat $out
