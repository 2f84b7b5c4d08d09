#!/bin/bash
# is a Go toolchain (or a network to fetch one) reachable from the GPU lease?
# Output: gpurun_out/r04/go_probe.txt (copied to profiles/r04/go_probe.txt).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
{
echo "== date"; date -u
echo "== which go / gccgo"; which go gccgo go1.21 2>&1; go version 2>&1
echo "== find go binaries"; find / -xdev \( -name 'go' -o -name 'gccgo*' -o -name 'gofmt' \) -type f 2>/dev/null | head
ls -d /usr/local/go /usr/lib/go* /opt/go* /root/go /root/sdk 2>&1
echo "== network"; timeout 8 curl -sI https://go.dev/dl/ 2>&1 | head -1; echo "curl rc=$?"
timeout 8 curl -sI https://proxy.golang.org 2>&1 | head -1; echo "curl rc=$?"
timeout 8 getent hosts go.dev; echo "getent rc=$?"
echo "== pip download"; timeout 20 pip download golang -d /tmp/x 2>&1 | tail -2
echo "== apt"; timeout 15 apt-get -s install golang-go 2>&1 | tail -3
echo "== conda / snap"; which conda snap 2>&1
echo "== /root/reference on box?"; ls /root/reference 2>&1 | head -3
echo "== nproc"; nproc; rocm-smi --showproductname 2>&1 | head -8
} > gpurun_out/r04/go_probe.txt 2>&1
cat gpurun_out/r04/go_probe.txt
timeout 300 python bench.py > gpurun_out/r04/bench_start.json 2> gpurun_out/r04/bench_start.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r04/bench_start.json
