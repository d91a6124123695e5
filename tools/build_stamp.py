"""Identity of the build the evidence under profiles/ describes.

  python tools/build_stamp.py            -> writes csts_amd/BUILD_STAMP.json (run in the build container, where .git exists,
                                            right before the GPU call that collects the evidence; the file travels with the snapshot)
  build_stamp.current()                  -> {"lib_sha256", "lib_f16_sha256", "source_commit", "source_dirty"}: the hashes are always
                                            recomputed from the .so files in the tree; the commit comes from git when there is one,
                                            else from BUILD_STAMP.json -- and only if that file's library hash matches the tree's.
The library build is reproducible (same sources + same toolchain -> same bytes), so lib_sha256 identifies the kernels that ran.
"""
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP = os.path.join(ROOT, "csts_amd", "BUILD_STAMP.json")


def _sha(path):
    if not os.path.exists(path):
        return None
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _git(*a):
    try:
        return subprocess.check_output(("git", "-C", ROOT) + a, stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def current():
    d = {"lib_sha256": _sha(os.path.join(ROOT, "csts_amd", "libcsts_hip.so")),
         "lib_f16_sha256": _sha(os.path.join(ROOT, "csts_amd", "libcsts_hip_f16.so"))}
    head = _git("rev-parse", "HEAD")
    if head:
        d["source_commit"] = head
        d["source_dirty"] = bool(_git("status", "--porcelain", "--untracked-files=no"))
    else:
        try:
            s = json.load(open(STAMP))
            if s.get("lib_sha256") == d["lib_sha256"]:
                d["source_commit"], d["source_dirty"] = s.get("source_commit"), s.get("source_dirty")
        except Exception:
            pass
    return d


if __name__ == "__main__":
    d = current()
    json.dump(d, open(STAMP, "w"), indent=1)
    print(json.dumps(d))
