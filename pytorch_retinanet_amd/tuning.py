"""Conv-stack tuning state for MI355X.

The ResNet/FPN/head convolutions run through MIOpen.  Measured on MI355X (R50-FPN, bf16,
batch 8 @800x1344, fwd+bwd+SGD; tools/conv_cfg.py):

    channels_last, heuristic pick (benchmark off)   1053 ms/step   (bwd-weight picks are pathological)
    NCHW,          heuristic pick                     89 ms/step
    NCHW,          exhaustive find (benchmark on)     65 ms/step   (342 s of find on a cold box)
    channels_last, exhaustive find                    51 ms/step   (209 s of find on a cold box)

so the framework runs channels_last + ``cudnn.benchmark`` and ships the find results for the
headline shapes (``miopen_db/*.ufdb.txt``, gfx950 / 256 CUs) so a cold box skips the search.
``use_shipped_miopen_db()`` must run before the first convolution of the process.
"""
import hashlib
import os
import shutil
import stat
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))


def use_shipped_miopen_db(rank: int = 0) -> str:
    """Point MIOpen's user find-db at a private, writable copy of the shipped one (one per rank:
    MIOpen rewrites the file).  Returns the directory.  No-op if the user already set the path."""
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = os.path.join(_HERE, "miopen_db")
    names = sorted(os.listdir(src))
    # keyed by the shipped files' content: a refreshed db gets a fresh directory instead of a stale copy
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(src, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    dst = os.path.join(tempfile.gettempdir(), f"retinanet_miopen_db_{os.getuid()}_{h.hexdigest()[:12]}_{rank}")
    os.makedirs(dst, mode=0o700, exist_ok=True)
    st = os.lstat(dst)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():     # somebody else's directory (or a symlink): do not use it
        dst = tempfile.mkdtemp(prefix="retinanet_miopen_db_")
    for name in names:
        target = os.path.join(dst, name)
        if not os.path.exists(target):
            shutil.copy(os.path.join(src, name), target)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst


def enable_conv_autotune() -> None:
    import torch
    torch.backends.cudnn.benchmark = True
