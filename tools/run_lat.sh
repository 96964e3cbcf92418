# forward time (hipGraph replay) of the folded 608 x 608 network at batch 1 / 2 / 4 with and without the latency form
for cfg in "MGD_LATENCY=0" "MGD_LATENCY=1"; do echo "== $cfg"; for b in 1 2 4; do env $cfg timeout -k 10 120 python tools/trace_timeline.py run $b graph 2>&1 | grep forward; done; done
