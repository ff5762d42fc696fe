# post-impact 10.3 M adaptive DFSPH (20 divergence iterations per step): loop tail against gated launches, strict and tolerance arithmetic
set -u
for T in 0 1; do for v in "" "SPHX_DFSPH_NO_TAIL=1" "SPHX_DFSPH_WINDOW=20"; do echo "== TOL=$T $v"; env TOL=$T $v timeout 400 python tools/postimpact_probe.py 190 300 2>&1 | grep -v "^PBD" | head -12; done; done
