"""Build libmtts_hip.so (all HIP kernels + the C-ABI) for gfx950 with hipcc, in-tree.

    python -m multilingual_text_to_speech_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached under csrc/build/ keyed by source mtime.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmtts_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result",
         "-ffp-contract=fast"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def _compile(src, obj):
    extra = os.environ.get("MTTS_EXTRA_FLAGS", "").split()          # e.g. -DMTTS_NO_STEP_PRIO for A/B builds
    cmd = [_hipcc(), *FLAGS, *extra, "-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def _flags_stamp(bdir):
    """The object cache is only valid for the flags it was built with: a change of FLAGS / MTTS_EXTRA_FLAGS (A/B builds with
    -D switches) rebuilds everything instead of silently linking stale objects."""
    import hashlib
    want = hashlib.sha256(" ".join(FLAGS + os.environ.get("MTTS_EXTRA_FLAGS", "").split()).encode()).hexdigest()
    path = os.path.join(bdir, "flags.sha256")
    have = open(path).read().strip() if os.path.exists(path) else None
    return want, have, path


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    # everything a translation unit may include: headers AND generated instruction streams (gemm_pipe_body.inc)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    newest_hdr = max([os.path.getmtime(h) for h in hdrs] + [0.0])
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    want, have, stamp = _flags_stamp(bdir)
    if want != have:
        force = True
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(bdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_hdr):
            jobs.append((s, o))
    if jobs:
        if verbose:
            print(f"[mtts build] compiling {len(jobs)} file(s) for gfx950", file=sys.stderr)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(lambda so: _compile(*so), jobs))
    with open(stamp, "w") as f:
        f.write(want)
    if jobs or not os.path.exists(OUT):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
