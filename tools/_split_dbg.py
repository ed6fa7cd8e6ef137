import sys, subprocess
code = r'''
import sys, torch
sys.path[:0]=["/root/repo","/root/repo/ubisoft-laforge-zeroeggs_amd"]
from zeggs import ops
M,N,K,lda,ldb,np_ = [int(x) for x in sys.argv[1:7]]
dev=torch.device("cuda:0")
torch.manual_seed(1)
A=torch.randn(K,lda,device=dev); B=torch.randn(K,ldb,device=dev); C=torch.zeros(M,N,device=dev)
ops.set_option("gemm_split_bf16", np_)
ops.gemm(A,B,C,M,N,K,(1,lda),(ldb,1),(N,1)); torch.cuda.synchronize()
ref=A[:,:M].double().t()@B[:,:N].double()
print("ok", M,N,K,np_, float((C.double()-ref).abs().max()/ref.pow(2).mean().sqrt()))
'''
open("/tmp/split_one.py","w").write(code)
for args in ["256 128 64 256 128 6", "256 128 256 256 128 6", "512 256 1024 512 256 6", "3072 1024 8160 3072 1024 6", "256 128 64 256 128 9", "256 128 64 256 128 3"]:
    r = subprocess.run([sys.executable, "/tmp/split_one.py"] + args.split(), capture_output=True, text=True)
    print(args, "->", (r.stdout.strip().splitlines() or ["-"])[-1], "|", (r.stderr.strip().splitlines() or ["-"])[-1][:150], flush=True)
