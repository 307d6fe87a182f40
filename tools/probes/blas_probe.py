import torch, time
dev="cuda"
def t(name, M,N,K, trans=False):
    a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)
    for _ in range(3): c=a@w.t()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c=a@w.t()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/20
    print(f"rocBLAS/hipBLASLt {name} M={M} N={N} K={K}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
torch.backends.cuda.matmul.allow_tf32=False
t("NT",82944,256,128); t("NT",82944,512,84); t("NT",124416,128,128); t("NT",131072,128,1024); t("NT",82944,128,84); t("NT", 8192,8192,8192)
# dW-like: [N,R]x[R,K]
M,N,K=82944,256,128
dy=torch.randn(M,N,device=dev); x=torch.randn(M,K,device=dev)
for _ in range(3): g=dy.t()@x
torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g=dy.t()@x
e1.record(); torch.cuda.synchronize(); us=e0.elapsed_time(e1)*1e3/20
print(f"dW R={M} N={N} K={K}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
