set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" "SPHX_NO_GRAPH=1" "SPHX_DFSPH_NO_TAIL=1"; do
  echo "== dfsph 250 50 under rocprofv3, $v"
  rm -rf /tmp/tr; env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python -X faulthandler $R/tools/r06_refscene_trace.py dfsph 250 50 > /tmp/tr.log 2>&1; echo "rc=$?"; grep -E "ms/step|Segmentation|File \"|line [0-9]+ in" /tmp/tr.log | head -12
done
echo "== without rocprofv3"; python -X faulthandler $R/tools/r06_refscene_trace.py dfsph 250 50 2>&1 | tail -2
echo "== pbd 250 50, graph off, under rocprofv3"
rm -rf /tmp/tr; SPHX_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python -X faulthandler $R/tools/r06_refscene_trace.py pbd 250 50 > /tmp/tr.log 2>&1; echo "rc=$?"; grep -E "ms/step|Segmentation|File \"|line [0-9]+ in" /tmp/tr.log | head -12
