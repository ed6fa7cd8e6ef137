"""Streaming gesture generation: audio arrives in chunks, pose frames leave as soon as their inputs exist.

Not in the reference (its generate_gesture() needs the whole wav); this is the chunked form of exactly the same
computation, so that a stream fed with any chunking produces the frames of the offline call:

  * mel + energy at the animation rate: STFT frames are local (n_fft 800 / hop 200, reflect padding only at the two
    ends of the signal); animation frame k interpolates the STFT frames around t = (80/60) k
    -> zeggs_mel_features_range over the samples received so far (csrc/mel.hip);
  * speech encoder: the k=31 convolution needs 15 frames of look-ahead (and replicate padding at the stream ends):
    each chunk is encoded over a window widened by 15 frames on both sides and the halo outputs are dropped;
  * decoder: zeggs_decoder_fwd_state resumes the autoregressive rollout from the GRU state / last pose of the previous
    chunk (2-slot rings, GEMV stage kernels at B = 1).

Latency = 15 frames of look-ahead (250 ms at 60 fps) + one STFT window; `finish()` flushes the tail with the true
right-edge rules.  BS.1770 loudness normalisation needs the whole signal and is therefore a pre-pass the caller
applies (or skips) before streaming, as in the offline path.
"""
import ctypes as C

import numpy as np
import torch

from . import audio, ops

LOOKAHEAD = 15        # (31 - 1) / 2 frames of the speech encoder's second convolution


class GestureStream:
    """speech_net / decoder: zeggs.modules instances on the device (eval mode); `first_pose`: the 16-tuple of
    anim.preprocess_animation (frame 0 is used); `style`: [1, S] tensor (constant style) ; `stats`: dict with
    audio_input_mean/std, anim_input_mean/std, anim_output_mean/std tensors; `audio_conf`: data_pipeline_conf
    ["audio_conf"] dict."""

    def __init__(self, speech_net, decoder, first_pose, style, stats, audio_conf, dt, feature_type=("mel_spec", "energy"),
                 fps=60.0, device="cuda"):
        g = audio_conf
        if g.get("normalize_loudness"):
            raise ValueError("loudness normalisation needs the whole signal: apply audio.normalize_loudness() first")
        if tuple(feature_type) != ("mel_spec", "energy"):
            raise NotImplementedError("streaming supports the shipped feature set [mel_spec, energy]")
        if g.get("resample_method", "linear") == "cubic":
            raise ValueError("resample_method 'cubic' is a spline over the whole signal: not available while the signal is still arriving")
        self.dev = torch.device(device)
        self.speech_net, self.decoder = speech_net.eval(), decoder.eval()
        self.stats = {k: v.to(self.dev, torch.float32) for k, v in stats.items()}
        self.dt, self.fps, self.fs = float(dt), float(fps), int(g["sampling_rate"])
        self.fb, min_clip = audio.mel_tables(g["filter_length"], self.fs, g["n_mel_channels"], g["mel_fmin"], g["mel_fmax"],
                                             g["min_clipping"], g["normalize_mel_bins"], g.get("real_amplitude", True), self.dev)
        self.mel = audio.MelDims(g["filter_length"], g["hop_length"], g["n_mel_channels"], self.fs, self.fps, float(min_clip),
                                 float(g.get("pre_emph_coeff", 0.97)) if g.get("pre_emphasis") else 0.0,
                                 audio.mel_flags(g.get("centered", True), g.get("normalize_range", True), g.get("resample_method", "linear")))
        f32 = lambda a: a[0:1].to(self.dev, torch.float32).contiguous()  # noqa: E731
        root_pos, root_rot, root_vel, root_vrt, lpos, lrot, ltxy, lvel, lvrt = first_pose[:9]
        self.gaze = f32(first_pose[14])                                   # [1, 3] constant gaze target
        self.style = style.to(self.dev, torch.float32).reshape(1, -1).contiguous()
        self.pose = torch.cat([f32(x).reshape(1, -1) for x in (root_vel, root_vrt, lpos, ltxy, lvel, lvrt)], dim=1)
        self.rpos, self.rrot = f32(root_pos), f32(root_rot)
        self.h = None                                                     # GRU state after the last emitted frame
        self.wav = np.zeros(0, np.float32)
        self.feats = torch.zeros(0, self.mel.n_mels + 1, device=self.dev)  # audio features of frames [0, n_feat)
        self.n_emitted = 0                                                # pose frames handed out so far (incl. frame 0)
        self.finished = False
        # give-up word of THIS stream's decoder calls (ZeggsDecCall.status): the state a chunk returns is the input of the next
        # one, so a rollout that did not complete must be noticed before it is carried forward (checked after every chunk)
        self.status = ops.new_status(self.dev)
        self.redone_chunks = 0
        self._L = ops.lib()
        self._L.zeggs_mel_frames_ready.restype = C.c_long
        self._L.zeggs_mel_range_workspace_bytes.restype = C.c_size_t

    # ------------------------------------------------------------------ audio features
    def _extend_features(self, k1, final):
        k0 = self.feats.shape[0]
        if k1 <= k0:
            return
        w = torch.as_tensor(self.wav, device=self.dev)
        L, d = self._L, self.mel
        ws = torch.empty(int(L.zeggs_mel_range_workspace_bytes(C.byref(d), C.c_long(k0), C.c_long(k1))), dtype=torch.uint8,
                         device=self.dev)
        out = torch.empty(k1 - k0, d.n_mels + 1, device=self.dev, dtype=torch.float32)
        rc = L.zeggs_mel_features_range(C.byref(d), C.c_void_p(w.data_ptr()), C.c_long(w.numel()), int(final),
                                        C.c_void_p(self.fb.data_ptr()), C.c_long(k0), C.c_long(k1),
                                        C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError("zeggs_mel_features_range: " + L.zeggs_last_error().decode())
        self.feats = torch.cat([self.feats, out], dim=0)

    # ------------------------------------------------------------------ decode frames [n_emitted, upto)
    def _decode(self, upto, n_total):
        """emit pose frames n_emitted .. upto-1; speech features exist for [0, feats.shape[0]); n_total = total frame
        count if known (stream end) else None"""
        k0, k1 = max(self.n_emitted, 1), upto
        out = {}
        if self.n_emitted == 0 and upto >= 1:
            out = dict(pose=[self.pose.clone()], rpos=[self.rpos.clone()], rrot=[self.rrot.clone()])
        if k1 <= k0:
            self.n_emitted = max(self.n_emitted, min(upto, 1))
            return self._pack(out)
        # speech encoding of frames k0-1 .. k1-1 over a window with 15-frame halos (true ends: replicate padding)
        lo = max(k0 - 1 - LOOKAHEAD, 0)
        hi = min(k1 + LOOKAHEAD, self.feats.shape[0]) if n_total is None else min(k1 + LOOKAHEAD, n_total)
        if n_total is None:
            assert hi == k1 + LOOKAHEAD, "not enough look-ahead features"
        x = ((self.feats[lo:hi] - self.stats["audio_input_mean"]) / self.stats["audio_input_std"])[None].contiguous()
        with torch.no_grad():
            sp = self.speech_net(x)[:, (k0 - 1 - lo):(k1 - lo)].contiguous()          # [1, N+1, SP]
        N1 = sp.shape[1]
        d = ops.DecDims(1, N1, self.pose.shape[1] + 3, self.pose.shape[1], sp.shape[2], self.style.shape[1],
                        self.decoder.recurrent_decoder.layer1.hidden_size, self.dt,
                        1 if hasattr(self.decoder.recurrent_decoder, "gammas_predictor") else 0)
        L = self._L
        params = [t.detach().to(torch.float32).contiguous() for t in ops.decoder_param_list(self.decoder)]
        P = ops._ptrs(ops.DecPtrs, ops.DEC_FIELDS, params)
        S = ops._ptrs(ops.DecStats, ("in_mean", "in_std", "out_mean", "out_std"),
                      [self.stats[k].contiguous() for k in ("anim_input_mean", "anim_input_std", "anim_output_mean",
                                                            "anim_output_std")])
        ws = torch.empty(int(L.zeggs_decoder_workspace_bytes(C.byref(d), 0)), dtype=torch.uint8, device=self.dev)
        gaze = self.gaze.repeat(N1, 1)[None].contiguous()
        style = self.style.repeat(N1, 1)[None].contiguous()
        pose = torch.empty(1, N1, d.PO, device=self.dev)
        rpos, rrot = torch.empty(1, N1, 3, device=self.dev), torch.empty(1, N1, 4, device=self.dev)
        h_out = torch.empty(2, 1, d.H, device=self.dev)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        call = ops.DecCall(0, 0, None, self.status.data_ptr(), 0)

        def run():
            rc = L.zeggs_decoder_fwd_state_ex(C.byref(d), C.byref(P), C.byref(S), p(self.pose), p(self.rpos), p(self.rrot),
                                              p(gaze), p(sp), p(style), p(pose), p(rpos), p(rrot), p(self.h), p(h_out), p(ws),
                                              C.c_size_t(ws.numel()), C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                              C.byref(call))
            if rc != 0:
                raise RuntimeError("zeggs_decoder_fwd_state_ex: " + L.zeggs_last_error().decode())
        run()
        if ops._persistent_live(0):       # a validated persistent kernel may have run: look at the word before the chunk's
            bits = int(self.status[0].item())     # state becomes the next chunk's input (the frames go to the host anyway)
            if bits:
                ops._warn_gave_up(bits, "the chunk")
                ops.set_option("persistent", 0)
                ops.fill_(self.status.view(torch.float32))
                self.redone_chunks += 1
                run()                     # same inputs (self.pose / self.h are untouched so far), stage launches
        self.h = h_out
        self.pose, self.rpos, self.rrot = pose[:, -1].contiguous(), rpos[:, -1].contiguous(), rrot[:, -1].contiguous()
        out.setdefault("pose", []).append(pose[0, 1:])
        out.setdefault("rpos", []).append(rpos[0, 1:])
        out.setdefault("rrot", []).append(rrot[0, 1:])
        self.n_emitted = k1
        return self._pack(out)

    @staticmethod
    def _pack(out):
        return {k: torch.cat(v, dim=0) for k, v in out.items()} if out else {}

    # ------------------------------------------------------------------ public API
    def push(self, wav_chunk):
        """append samples (float32 in [-1, 1)); returns {"pose": [n, PO], "rpos": [n, 3], "rrot": [n, 4]} for the n new
        frames that became computable (possibly an empty dict)"""
        assert not self.finished
        self.wav = np.concatenate([self.wav, np.asarray(wav_chunk, np.float32)])
        ready = int(self._L.zeggs_mel_frames_ready(C.byref(self.mel), C.c_long(len(self.wav))))
        # never run ahead of the final frame count (it can only grow with more samples)
        ready = min(ready, audio.n_anim_frames(len(self.wav)) - 1)
        self._extend_features(ready, final=False)
        upto = self.feats.shape[0] - LOOKAHEAD
        if upto <= self.n_emitted:
            return {}
        return self._decode(upto, None)

    def finish(self):
        """end of the signal: flush the remaining frames (right-edge reflect / replicate rules of the offline path)"""
        assert not self.finished
        self.finished = True
        n_total = audio.n_anim_frames(len(self.wav))
        self._extend_features(n_total, final=True)
        return self._decode(n_total, n_total)

    @staticmethod
    def split_pose(pose, J):
        """pose rows [n, 6+15J] -> root_vel, root_vrt, lpos, ltxy, lvel, lvrt (reference output-vector order)"""
        n = pose.shape[0]
        o = [6, 6 + 3 * J, 6 + 9 * J, 6 + 12 * J]
        return (pose[:, 0:3], pose[:, 3:6], pose[:, 6:o[1]].reshape(n, J, 3), pose[:, o[1]:o[2]].reshape(n, J, 2, 3),
                pose[:, o[2]:o[3]].reshape(n, J, 3), pose[:, o[3]:].reshape(n, J, 3))
