"""One process per GPU: the rank arithmetic and the spawner behind `bench.py --gpus N`.

The reference has no counterpart -- main.go runs one Decoder on one rtl_tcp stream.  A multi-GPU host owns one rank per
device; this module decides, from `--gpus`, the environment (torchrun's RANK / LOCAL_RANK / WORLD_SIZE, if any) and the
number of gfx950 devices, whether the current process IS a rank, has to START the ranks itself, or must refuse -- a run
that asked for N ranks never silently becomes a run of one.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time
from typing import Dict, List, Mapping, Optional, Sequence, Tuple


class LaunchError(RuntimeError):
    """`--gpus` cannot be honoured; str(e) is the one-line reason."""


def free_port() -> int:
    """A TCP port nobody listens on right now (127.0.0.1)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_plan(gpus: int, env: Mapping[str, str], n_devices: int, port: Optional[int] = None) -> Tuple[str, List[Dict[str, str]]]:
    """What `--gpus gpus` means in this process.

    -> ("rank", [])      the process is a rank already: a launcher (torchrun) set WORLD_SIZE == gpus, or gpus == 1;
       ("spawn", envs)   no launcher: start `gpus` processes, envs[r] = the variables rank r gets on top of `env`
                         (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT).
    Raises LaunchError when the request cannot be met: --gpus != WORLD_SIZE, fewer devices than ranks, a LOCAL_RANK
    without a device."""
    if gpus < 1:
        raise LaunchError(f"--gpus {gpus}: at least one rank")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise LaunchError(f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")
        local = int(env.get("LOCAL_RANK", env.get("RANK", "0")))
        rank = int(env.get("RANK", "0"))
        if not 0 <= rank < world:
            raise LaunchError(f"RANK={rank} outside WORLD_SIZE={world}")
        if local >= n_devices:
            raise LaunchError(f"LOCAL_RANK={local} but {n_devices} device(s) visible")
        return "rank", []
    if gpus == 1:
        if n_devices < 1:
            raise LaunchError("1 rank requested, 0 devices")
        return "rank", []
    if n_devices < gpus:
        raise LaunchError(f"{gpus} ranks requested, {n_devices} device{'s' if n_devices != 1 else ''}")
    port = free_port() if port is None else port
    envs = []
    for r in range(gpus):
        envs.append({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(gpus), "LOCAL_WORLD_SIZE": str(gpus),
                     "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                     "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    return "spawn", envs


def spawn_ranks(argv: Sequence[str], envs: Sequence[Mapping[str, str]], base_env: Optional[Mapping[str, str]] = None,
                poll_s: float = 0.05) -> int:
    """Start one process per entry of `envs` (argv identical, environment = base_env + envs[r]), wait for all of them and
    return the job's exit code: 0 when every rank exited 0, otherwise the code of the FIRST rank seen failing (a negative
    Popen code = killed by a signal -> 128 + signal).  Rank 0 inherits stdout (its single JSON line is the job's output);
    the other ranks' stdout goes to stderr.  When a rank fails, the others get SIGTERM (by PID) so that nobody waits in a
    collective forever; the exit status of a rank this function terminated itself (-SIGTERM, or whatever its handler
    turned that into) says nothing about the job and is not folded in -- bench.py's refusal / mismatch codes (2, 3 ... 7)
    reach the caller as they are."""
    base = dict(os.environ if base_env is None else base_env)
    procs: List[subprocess.Popen] = []
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen(list(argv), env={**base, **e}, stdout=None if r == 0 else sys.stderr))
    rc = 0
    terminated = set()      # ranks that were sent SIGTERM by us
    try:
        live = set(range(len(procs)))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code == 0 or r in terminated:
                    continue
                if rc == 0:
                    rc = code if code > 0 else 128 - code
                for o in sorted(live - terminated):     # a failed rank: the others would hang in the next collective
                    procs[o].send_signal(signal.SIGTERM)
                    terminated.add(o)
            if live:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc

