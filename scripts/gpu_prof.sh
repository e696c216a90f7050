#!/bin/bash
# usage: gpu_prof.sh <name> <command...>: rocprofv3 kernel-trace stats of <command>, csv under gpurun_out/<name>/,
# prints the top kernels (calls, average ns, total ns)
NAME="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p "$ROOT/gpurun_out/$NAME"
export TMPDIR=/tmp; export PYTHONPATH="$ROOT:$PYTHONPATH"
# rocprofv3 runs from /tmp (its scratch files), so relative script paths are made absolute
ARGS=(); for a in "$@"; do if [ -e "$ROOT/$a" ]; then ARGS+=("$ROOT/$a"); else ARGS+=("$a"); fi; done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/$NAME" -o "$NAME" -- "${ARGS[@]}" > "$ROOT/gpurun_out/$NAME/stdout.log" 2>&1
cd "$ROOT"
F=$(find "gpurun_out/$NAME" -name '*kernel_stats.csv' | head -1)
python - "$F" <<'EOF'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
for r in rows[:28]:
    print('%-110s calls %5s avg_us %9.1f total_ms %8.2f' % (r[0][:110], r[1], float(r[3]) / 1e3, float(r[2]) / 1e6))
EOF
find "gpurun_out/$NAME" -name '*kernel_trace.csv' -size +20M -delete
