# the 1x1 (+ skip) ResidualBlock layers with pieces removed (DIAG library, results are WRONG on purpose), next to plain
# streaming ops of the same byte counts.  Needs: make -C shallow-ntc_amd/csrc DIAG=1 BUILD=build_diag LIB=../lib/libsntc_hip_diag.so
R=${GRAFT_REPO_ROOT:-.}; cd $R
export SNTC_LIB=$R/shallow-ntc_amd/lib/libsntc_hip_diag.so
L="python3 tools/one_layer.py --kind conv --k 1 --s 1 --n 18 --hw 256 384 --reps 8"
for dbg in 0 16 32 48 64 1 3 67 83 115; do
  echo "SNTC_GG_DBG=$dbg  96->192 + skip"; SNTC_GG_DBG=$dbg $L --cin 96 --cout 192 --epi 2>/dev/null | cut -c1-150
done
for dbg in 0 16 64 3 80; do
  echo "SNTC_GG_DBG=$dbg  192->96"; SNTC_GG_DBG=$dbg $L --cin 192 --cout 96 2>/dev/null | cut -c1-150
done
python3 - <<'PY'
import torch
dev = torch.device("cuda:0")
n = 18 * 256 * 384
a = torch.randn((n, 192), device=dev); b = torch.randn((n, 192), device=dev); y = torch.empty_like(a); x = torch.randn((n, 96), device=dev)
def t(fn, bytes_):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return f"{ms:.4f} ms  {bytes_ / ms / 1e9:.2f} TB/s"
print("copy 1.36 GB -> 1.36 GB      ", t(lambda: y.copy_(a), 2 * a.numel() * 4))
print("add  2 x 1.36 GB -> 1.36 GB  ", t(lambda: torch.add(a, b, out=y), 3 * a.numel() * 4))
print("read 0.68 GB (sum)            ", t(lambda: x.sum(), x.numel() * 4))
PY
