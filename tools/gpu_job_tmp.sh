cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf /tmp/trk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trk -o b -- python tools/mode_trace.py track 40 --repeat 2 > /dev/null 2>&1
f=$(find /tmp/trk -name "b_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print('%-46s calls %6s avg %8.1f us  %5.2f %%' % (r['Name'].split('(')[0].replace('void ', '')[:46], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
timeout 900 python -m pytest tests/test_parity_at_size_configs.py -m gpu -x -q -k "lookahead and 0.03" 2>&1 | grep -E "^E |assert" | head -12
