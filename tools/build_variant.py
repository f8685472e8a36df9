#!/usr/bin/env python3
"""
Build a variant of the library with one source recompiled under extra -D flags:
    python tools/build_variant.py <name> <source.hip> -DLK_ALS_RING=8 ...
-> tools/_variants/lkamd_<name>.so (the other objects come from lkpy_amd/csrc/_obj, so run
`python -m lkpy_amd.csrc.build` first).  tools/als_variants.py times such builds side by side.
"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lkpy_amd.csrc import build as B  # noqa: E402


def main():
    name, src, *flags = sys.argv[1:]
    src = (B.HERE / src).resolve()
    out = ROOT / "tools" / "_variants"
    out.mkdir(exist_ok=True)
    obj = out / f"{src.stem}_{name}.o"
    subprocess.check_call([B.HIPCC, *B.FLAGS, *flags, "-c", str(src), "-o", str(obj)])
    objs = [obj if o.stem == src.stem else o for o in sorted(B.OBJ.glob("*.o"))]
    so = out / f"lkamd_{name}.so"
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(so),
                           *map(str, objs)])
    print(so)


if __name__ == "__main__":
    main()
