import sys, time
sys.path[:0]=["/root/repo","/root/repo/ubisoft-laforge-zeroeggs_amd"]
import torch, numpy as np, bench
from zeggs import ops, synth, modules
dev=torch.device("cuda:0")
torch.manual_seed(1234)
de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, 2).to(dev).eval()
T=108000
args = bench.decode_args(de, dev, T)
# args: (de, pose0, rpos0, rrot0, gaze, speech, style, in_mean, in_std, out_mean, out_std, dt)
def sync(): torch.cuda.synchronize()
with torch.no_grad():
    ops.decoder_core(*args); sync()
    t0=time.perf_counter(); ops.decoder_core(*args); sync(); print("one launch: %.1f ms"%((time.perf_counter()-t0)*1e3))
    _, pose0, rpos0, rrot0, gaze, speech, style, im, isd, om, osd, dt = args
    for chunk in (8192, 16384, 32768):
        for rep in range(2):
            sync(); t0=time.perf_counter()
            state=(pose0, rpos0, rrot0, None); k=0
            while k < T-1:
                n=min(chunk, T-1-k)
                p,rp,rr,h = ops.decoder_chunk(de, state[0], state[1], state[2], gaze[:, k:k+n+1], speech[:, k:k+n+1], style[:, k:k+n+1], im, isd, om, osd, dt, h_in=state[3])
                state=(p[:,-1], rp[:,-1], rr[:,-1], h); k+=n
            sync(); print("chunk %d: %.1f ms"%(chunk,(time.perf_counter()-t0)*1e3))
    # the conversion of one chunk's frames to rows of the BVH motion block (runs between two decode launches: the persistent
    # kernel leaves no room beside it)
    import ctypes as C
    from zeggs import anim
    J = len(synth.PARENTS)
    n = 8192
    p, rp, rr, h = ops.decoder_chunk(de, pose0, rpos0, rrot0, gaze[:, :n + 1], speech[:, :n + 1], style[:, :n + 1], im, isd, om, osd, dt)
    d = anim.BvhDims(0, J, 1); d.T = n
    d.start_pos[:] = [0.0, 0.0, 0.0]; d.start_rot[:] = [1.0, 0.0, 0.0, 0.0]
    P = p[0, 1:]
    a_rpos, a_rrot = rp[0, 1:].contiguous(), rr[0, 1:].contiguous()
    a_lpos, a_ltxy = P[:, 6:6 + 3 * J].contiguous(), P[:, 6 + 3 * J:6 + 9 * J].contiguous()
    _, seq = anim.bvh_header(np.zeros((J, 3)), synth.PARENTS, [f"j{i}" for i in range(J)], "zyx", n, dt)
    seq_dev = torch.as_tensor(np.asarray(seq, np.int32), device=dev)
    table = torch.empty(n, 3 + 3 * J, dtype=torch.float64, device=dev)
    ref_pos, ref_rot = rpos0.float().contiguous(), rrot0.float().contiguous()
    L = ops.lib()
    def conv():
        L.zeggs_pose_to_bvh_table(C.byref(d), C.c_void_p(a_rpos.data_ptr()), C.c_void_p(a_rrot.data_ptr()), C.c_void_p(a_lpos.data_ptr()),
                                  C.c_void_p(a_ltxy.data_ptr()), C.c_void_p(ref_pos.data_ptr()), C.c_void_p(ref_rot.data_ptr()),
                                  C.c_void_p(seq_dev.data_ptr()), C.c_void_p(table.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    conv(); sync(); t0 = time.perf_counter()
    for _ in range(10): conv()
    sync(); print("pose_to_bvh_table, 8192 frames: %.3f ms" % ((time.perf_counter() - t0) * 1e2))
