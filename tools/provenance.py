"""Provenance stamp for the profile-derived JSON tables under profiles/ (hbm_traffic*.json, valu_*.json): the hashes of
the kernel sources the counters were collected on, the git revision if known, the date and the command.  bench.py drops a
table's numbers when the sources of the kernel in question have changed since (bench.profile_value)."""
import datetime
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hashes():
    d = os.path.join(ROOT, "splatco_amd", "csrc")
    return {n: hashlib.sha256(open(os.path.join(d, n), "rb").read()).hexdigest()[:16]
            for n in sorted(os.listdir(d)) if n.endswith((".hip", ".h"))}


def stamp(command):
    git = os.environ.get("SPLATCO_GIT_SHA")          # the GPU box has no .git: tools/profile_*.sh pass it along when set
    if not git:
        try:
            git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except OSError:
            git = None
    lib = os.path.join(ROOT, "splatco_amd", "csrc", "libsplatco_raster.so")
    return {"sources": source_hashes(), "git": git, "date": datetime.datetime.now().strftime("%Y-%m-%d %H:%M"),
            "command": command,
            "library_sha256_16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None}
