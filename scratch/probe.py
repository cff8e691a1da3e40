import ctypes, torch, os, subprocess
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libprobe.so'))
print('hip runtime version', lib.probe_rtver(), 'torch hip', torch.version.hip)
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
K = 64
A = torch.randn(16, K, device='cuda'); B = torch.randn(K, 16, device='cuda'); D = torch.zeros(16, 16, device='cuda')
s = torch.cuda.current_stream().cuda_stream
rc = lib.probe_mfma(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(D.data_ptr()), K, ctypes.c_void_p(s))
torch.cuda.synchronize()
ref = (A.double() @ B.double()).float()
print('rc', rc, 'maxerr', (D - ref).abs().max().item())
# exact fmaf chain check on CPU
import numpy as np
a = A.cpu().numpy(); b = B.cpu().numpy(); out = np.zeros((16,16), np.float32)
acc = np.zeros((16,16), np.float64)
chain = np.zeros((16,16), np.float32)
for k in range(K):
    chain = np.float32(np.float64(a[:, k:k+1]) * np.float64(b[k:k+1, :]) + np.float64(chain))  # fma emulation via f64 (exact product, one rounding)
print('bitexact vs k-ordered fma chain:', bool((chain == D.cpu().numpy()).all()), np.abs(chain - D.cpu().numpy()).max())
# side stream test
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    D.zero_()
    lib.probe_mfma(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(D.data_ptr()), K, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
st.synchronize(); print('side stream maxerr', (D - ref).abs().max().item())
print('nproc', os.cpu_count()); print(subprocess.run('lscpu | head -20; free -g | head -2', shell=True, capture_output=True, text=True).stdout)
for l in open('/proc/self/maps'):
    if 'amdhip64' in l: print(l.strip().split()[-1]); break
