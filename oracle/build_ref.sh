#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the reference's *in-tree* C extension modules
# (base, blas, lapack, misc_solvers) from the sources where they lie under
# /root/reference, into oracle/_ref/ (git-ignored).  Nothing from the reference is
# copied into the repository; only compiled .so files land in oracle/_ref/.
#
# The reference's Cholesky arithmetic itself (SuiteSparse CHOLMOD 7.8.2,
# /root/reference/.ci/config/versions.env:7) is an un-vendored third-party library
# that is absent from this image, so src/C/cholmod.c is UNBUILDABLE here and is not
# attempted.  BLAS/LAPACK come from the OpenBLAS that ships inside scipy (symbols
# carry a scipy_ prefix, mapped below).
#
# Used by tests/golden/make_goldens.py (this container only; /root/reference does
# not exist on the GPU box).
set -euo pipefail
REF=${KVX_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
SRC="$REF/src/C"
[ -d "$SRC" ] || { echo "reference not present at $REF; skipping _ref build"; exit 0; }
mkdir -p "$OUT"
PYINC=$(python3 -c "import sysconfig;print(sysconfig.get_paths()['include'])")
EXT=$(python3 -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
OB=$(python3 - <<'EOF'
import glob, os, scipy
d = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
print(sorted(glob.glob(os.path.join(d, "libscipy_openblas*.so")))[0])
EOF
)
# symbol map: reference calls  xxx_  -> scipy's OpenBLAS exports scipy_xxx_
grep -E '^#define [a-z0-9_]+_ ' "$SRC/blas_redefines.h" | awk '{print "#define "$2" scipy_"$2}' | sort -u > "$OUT/rename.h"
for s in dlange_ zlange_ dlarfg_ dlarfx_ zlarfg_ zlarfx_; do echo "#define $s scipy_$s" >> "$OUT/rename.h"; done
CF="-O2 -fPIC -shared -w -I$PYINC -I$SRC -include $OUT/rename.h"
LD="$OB -lm -Wl,-rpath,$(dirname "$OB")"
gcc $CF -o "$OUT/base$EXT"         "$SRC/base.c" "$SRC/dense.c" "$SRC/sparse.c" $LD
gcc $CF -o "$OUT/blas$EXT"         "$SRC/blas.c"         $LD
gcc $CF -o "$OUT/lapack$EXT"       "$SRC/lapack.c"       $LD
gcc $CF -o "$OUT/misc_solvers$EXT" "$SRC/misc_solvers.c" $LD
echo "built reference in-tree extensions into $OUT"
