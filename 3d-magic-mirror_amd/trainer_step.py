"""Trainer-SHAPED step around the render path (BASELINE config 3: ellipsoid template, B=48, 256x256, ResNet-18 encoder on
PyTorch-ROCm): the call sequence of one generator iteration of /root/reference/trainer.py, with stock-PyTorch networks of
the reference's output shapes in place of its encoders.  It exists to exercise and time the hot path the way its caller
drives it -- four renders in the trainer's dependency order, one recon_data, the regularisers, ONE backward -- not to train
anything: the encoders, discriminator, data loading and the D update are out of scope and are NOT re-built here.

    Ae   = netE(Xa);                         Xer,   Ae   = render(**Ae)        trainer.py:273-276
    Ae90 = deep_copy(Ae), random azimuths;   Xer90, Ae90 = render(**Ae90)      :279-291,347      (--hard)
    Ai   = lerp of two permutations of Ae;   Xir,   Ai   = render(**Ai)        :305-345
    Aire = netE(Xir.detach());               _,     Aire = render(**Aire)      :365-367
    lossR = critic(Xer90, Xir) + lambda_data * recon_data(Xer, Xa) + regularization(Ae, Ai, Aire)     :429-507
    lossR.backward(); optimizerE.step()                                        :509-518

The encoder is a ResNet-18 (BasicBlock x [2,2,2,2]) with the reference's 4-channel stem and stride-1 last stage
(network/model_res.py:688-734 wraps torchvision's; torchvision is absent here, so the standard architecture is written out in
plain torch.nn with random initialisation), one trunk for shape and one for texture as in the reference's ShapeEncoder /
TextureEncoder, small conv stacks for camera / light / background (its Base_4C role).  The texture head emits a flow and goes
through ``sample_texture`` (the HIP texture-flow kernel, model_res.py:597-612).  A two-layer conv critic stands in for netD so
that, as in the reference, image gradient reaches the rasteriser through Xer90 and Xir as well.
"""
import math
import time
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from .diff_render import DiffRender, deep_copy
from .texture_flow import sample_texture


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.down is None else self.down(x)))


class ResNet18_4C(nn.Module):
    """ResNet-18 trunk, 4-channel 7x7/2 stem, last stage at stride 1: (B,4,H,W) -> (B,512,H/16,W/16)."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(4, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
        cfg = [(64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 1)]
        self.layers = nn.ModuleList([nn.Sequential(BasicBlock(a, b, s), BasicBlock(b, b, 1)) for a, b, s in cfg])

    def forward(self, x, taps=False):
        x = self.stem(x)
        feats = []
        for l in self.layers:
            x = l(x)
            feats.append(x)
        return feats if taps else x


def _stack(cin, widths):
    mods, c = [], cin
    for w in widths:
        mods += [nn.Conv2d(c, w, 3, 2, 1, bias=False), nn.BatchNorm2d(w), nn.ReLU()]
        c = w
    return nn.Sequential(*mods)


class AttributeNet(nn.Module):
    """Image (B,4,H,W) in [0,1] -> the attribute dict of networks.py:635-646 (value ranges of model_res.py:196-216,325-337,385-395)."""

    def __init__(self, vertices_init, elev=(0.0, 30.0), dist=(2.0, 7.0), azi_scope=360.0, bg=True):
        super().__init__()
        V = vertices_init.shape[0]
        self.register_buffer("vertices_init", vertices_init.reshape(1, V, 3).float())
        self.elev, self.dist, self.azi_scope, self.bg = elev, dist, azi_scope, bg
        self.shape_trunk = ResNet18_4C()
        self.shape_head = nn.Linear(512, V * 3)
        nn.init.normal_(self.shape_head.weight, std=1e-3); nn.init.zeros_(self.shape_head.bias)
        self.camera = _stack(4, [32, 64, 128, 256, 256])
        self.camera_head = nn.Linear(256, 6)
        self.light = _stack(4, [32, 64, 128, 128])
        self.light_head = nn.Linear(128, 9)
        self.tex_trunk = ResNet18_4C()
        self.tex_up = nn.ModuleList([nn.Sequential(nn.Conv2d(c, o, 3, 1, 1, bias=False), nn.BatchNorm2d(o), nn.ReLU())
                                     for c, o in ((512 + 256, 256), (256 + 128, 128), (128 + 64, 64), (64, 32))])
        self.tex_flow = nn.Conv2d(32, 2, 3, 1, 1)
        if bg:
            self.bg_net = nn.Sequential(nn.Conv2d(4, 32, 3, 1, 1), nn.ReLU(), nn.Conv2d(32, 32, 3, 1, 1), nn.ReLU(), nn.Conv2d(32, 3, 3, 1, 1))
        self.register_buffer("light_scale", torch.tensor([[0.5] + [0.1] * 8]))
        self.register_buffer("light_bias", torch.tensor([[3.0] + [0.0] * 8]))

    def forward(self, x):
        Bn, _, H, W = x.shape
        xn = (x - 0.5) * 2.0
        d = self.shape_head(F.adaptive_avg_pool2d(self.shape_trunk(xn), 1).flatten(1))
        delta = (0.5 * torch.tanh(d)).view(Bn, -1, 3)
        delta = delta - delta.mean(dim=1, keepdim=True)
        c = self.camera_head(F.adaptive_avg_pool2d(self.camera(xn), 1).flatten(1))
        distances = self.dist[0] + torch.sigmoid(c[:, 0]) * (self.dist[1] - self.dist[0])
        elevations = self.elev[0] + torch.sigmoid(c[:, 1]) * (self.elev[1] - self.elev[0])
        azimuths = -torch.atan2(c[:, 3], c[:, 2] + 1e-6) * (180.0 / math.pi) / 360.0 * self.azi_scope
        biases = torch.tanh(c[:, 4:6])
        lights = torch.tanh(self.light_head(F.adaptive_avg_pool2d(self.light(xn), 1).flatten(1))) * self.light_scale + self.light_bias
        f1, f2, f3, f4 = self.tex_trunk(xn, taps=True)            # 1/4 (64), 1/8 (128), 1/16 (256), 1/16 (512)
        t = self.tex_up[0](torch.cat([f4, f3], 1))
        t = self.tex_up[1](torch.cat([F.interpolate(t, size=f2.shape[2:], mode="nearest"), f2], 1))
        t = self.tex_up[2](torch.cat([F.interpolate(t, size=f1.shape[2:], mode="nearest"), f1], 1))
        t = self.tex_up[3](F.interpolate(t, size=(H // 2, W // 2), mode="nearest"))
        ys = torch.linspace(-1, 1, H, device=x.device).view(1, 1, H, 1)
        xs = torch.linspace(-1, 1, W, device=x.device).view(1, 1, 1, W)
        ident = torch.cat([xs.expand(1, 1, H, W), ys.expand(1, 1, H, W)], 1)
        flow = torch.tanh(F.interpolate(self.tex_flow(t), size=(H, W), mode="bilinear", align_corners=False) + ident)
        textures = sample_texture(x[:, :3], flow)                 # (B,3,2H,W): HIP bicubic texture-flow kernel + mirror
        return {"azimuths": azimuths, "elevations": elevations, "distances": distances, "biases": biases,
                "vertices": self.vertices_init + delta, "delta_vertices": delta, "textures": textures, "lights": lights,
                "img_feats": None, "bg": torch.sigmoid(self.bg_net(xn)) if self.bg else None}


def default_opt():
    """train.py:39-127 defaults, with the README's CUB flags (--bg --hard --chamfer ...)."""
    return types.SimpleNamespace(bg=True, hard=True, lambda_data=1.0, lambda_reg=0.1, lambda_flipz=0.1, lambda_ic=1.0, lambda_edge=0.001,
                                 lambda_depth=0.0, lambda_depthR=0.0, lambda_depthC=0.0, lambda_deform=0.1, lambda_gan=1e-4, ganw=1.0,
                                 lambda_contour=0.0, flipL1=False, L1=False, chamfer=True, azim=1.0, temp=2.0, bias_range=0.3,
                                 azi_scope=360.0, hard_range=0, lr=1e-4, beta1=0.5)


class TrainerStep:
    def __init__(self, template, image_size, batch, device, ratio=1, opt=None, seed=0, lean=False, many=False):
        """lean: the fourth render (trainer.py:367, whose image is discarded) as DiffRender.render_geometry -- same losses, same gradients.
        many (with lean): the three renders whose attributes exist up front (trainer.py:276,345,347) as ONE DiffRender.render_many call of 3B
        images.  Pays where a batch of 48 does not fill the chip (128x128); at 256x256 one pass over 144 images takes as long as three over 48
        and the concatenations cost more than the launches saved (profiles/r04_render_path.md)."""
        self.opt = opt or default_opt()
        self.dev, self.B = device, batch
        self.lean = bool(lean)                                   # render #4 geometry-only
        self.many = bool(many) and self.lean                     # renders #1-#3 as one render_many call
        self.dr = DiffRender(template, image_size, ratio=ratio)
        torch.manual_seed(seed)
        self.netE = AttributeNet(self.dr.vertices_init, bg=self.opt.bg).to(device)
        self.critic = nn.Sequential(nn.Conv2d(3, 32, 4, 2, 1), nn.LeakyReLU(0.2), nn.Conv2d(32, 64, 4, 2, 1), nn.LeakyReLU(0.2),
                                    nn.Conv2d(64, 1, 4, 2, 1)).to(device)
        for p in self.critic.parameters():
            p.requires_grad_(False)
        self.optimizerE = torch.optim.Adam(self.netE.parameters(), lr=self.opt.lr, betas=(self.opt.beta1, 0.999))
        H, W = self.dr.render_height, self.dr.image_size
        g = torch.Generator().manual_seed(seed + 1)
        rgb = torch.rand(batch, 3, H, W, generator=g)
        ys = (torch.arange(H).float() + 0.5 - H / 2.0) / (0.4 * H)
        xs = (torch.arange(W).float() + 0.5 - W / 2.0) / (0.35 * W)
        m = ((ys[:, None] ** 2 + xs[None, :] ** 2) <= 1.0).float().expand(batch, 1, H, W)
        self.Xa = torch.cat([rgb, m], 1).contiguous().to(device)
        self.gen = torch.Generator(device=device).manual_seed(seed + 2)
        self.last = {}

    def _render(self, slot, A):
        """render #slot of the iteration"""
        return self.dr.render(**A, no_mask=self.opt.bg)

    def _u(self, *shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, device=self.dev, generator=self.gen) * (hi - lo) + lo

    def step(self, optimize=True):
        o, dr, Bn = self.opt, self.dr, self.B
        self.optimizerE.zero_grad(set_to_none=True)
        Ae = self.netE(self.Xa)
        if not self.many:
            Xer, Ae = self._render(0, Ae)                                                  # render #1
        Ae90 = deep_copy(Ae)
        sign = torch.where(self._u(Bn) < 0.5, -1.0, 1.0)
        Ae90["azimuths"] = -self._u(Bn, lo=o.hard_range, hi=180.0 - o.hard_range) * sign
        ra, rb = torch.randperm(Bn, device=self.dev, generator=self.gen), torch.randperm(Bn, device=self.dev, generator=self.gen)
        Aa, Ab = deep_copy(Ae, ra), deep_copy(Ae, rb)
        a_s, a_t, a_l = self._u(Bn, 1, 1), self._u(Bn, 1, 1, 1), self._u(Bn, 1)
        Ai = {"azimuths": -self._u(Bn, lo=-o.azi_scope / 2, hi=o.azi_scope / 2),
              "elevations": self._u(Bn, lo=self.netE.elev[0], hi=self.netE.elev[1]),
              "distances": self._u(Bn, lo=self.netE.dist[0], hi=self.netE.dist[1]),
              "biases": self._u(Bn, 2, lo=-o.bias_range, hi=o.bias_range),
              "vertices": a_s * Aa["vertices"] + (1 - a_s) * Ab["vertices"],
              "delta_vertices": a_s * Aa["delta_vertices"] + (1 - a_s) * Ab["delta_vertices"],
              "textures": a_t * Aa["textures"] + (1 - a_t) * Ab["textures"],
              "bg": (a_t * Aa["bg"] + (1 - a_t) * Ab["bg"]) if o.bg else None,
              "lights": a_l * Aa["lights"] + (1 - a_l) * Ab["lights"]}
        if self.many and o.hard:                                                           # renders #1-#3 in one pass over 3B images
            (Xer, Ae), (Xir, Ai), (Xer90, Ae90) = dr.render_many([Ae, Ai, Ae90], no_mask=o.bg)
        elif self.many:
            (Xer, Ae), (Xir, Ai) = dr.render_many([Ae, Ai], no_mask=o.bg)
            Xer90, Ae90 = Xer, Ae
        else:
            Xir, Ai = self._render(1, Ai)                                                  # render #2
            Xer90, Ae90 = self._render(2, Ae90) if o.hard else (Xer, Ae)                   # render #3
        Aire = self.netE(Xir.detach().clone())
        if self.lean:
            Aire = dr.render_geometry(**Aire)                                              # render #4: its image is discarded (trainer.py:367)
        else:
            _, Aire = self._render(3, Aire)                                                # render #4 (face_normals only)
        outs = self.critic(torch.cat((Xer90[:, :3], Xir[:, :3]), 0))
        o1, o2 = torch.split(outs, Bn, 0)
        lossR_fake = o.lambda_gan * (-o1.mean() - o.ganw * o2.mean()) / (1.0 + o.ganw)
        lossR_data = o.lambda_data * dr.recon_data(Xer, self.Xa, no_mask=o.bg, contour=o.lambda_contour)
        lossR_reg, lossR_flip, lossR_IC = dr.regularization(Ae, Ai, Aire, o)
        lossR = lossR_fake + lossR_reg + lossR_flip + lossR_data + lossR_IC
        lossR.backward()
        if optimize:
            self.optimizerE.step()
        self.last = {"loss": lossR.detach(), "data": lossR_data.detach(), "reg": lossR_reg.detach(), "flip": lossR_flip.detach(),
                     "ic": lossR_IC.detach(), "fake": lossR_fake.detach()}
        return lossR.detach()

    def render_path_only(self):
        """The same four renders + recon_data + regularisers + backward on DETACHED attributes of the last netE output: the
        share of the step that is this repo's path (no encoder, no optimizer)."""
        o, dr = self.opt, self.dr
        with torch.no_grad():
            A0 = self.netE(self.Xa)
        def leaf(A):
            return {k: (v.detach().clone().requires_grad_(True) if torch.is_tensor(v) else v) for k, v in A.items()}
        sets = [leaf(A0) for _ in range(4)]                       # four leaf copies of the attributes, made ONCE: copying 300 MB of textures per run is
        def run():                                               # not part of the path that is being timed
            for A in sets:
                for v in A.values():
                    if torch.is_tensor(v):
                        v.grad = None
            Ae, Ai, A9, Ar = (dict(A) for A in sets)
            if self.many:
                (Xer, Ae), (Xir, Ai), (Xer90, A9) = dr.render_many([Ae, Ai, A9], no_mask=o.bg)
            else:
                Xer, Ae = self._render(0, Ae)
                Xir, Ai = self._render(1, Ai)
                Xer90, A9 = self._render(2, A9)
            if self.lean:
                Ar = dr.render_geometry(**Ar)
            else:
                _, Ar = self._render(3, Ar)
            l = dr.recon_data(Xer, self.Xa, no_mask=o.bg) + 1e-4 * (Xer90[:, :3].mean() + Xir[:, :3].mean())
            r1, r2, r3 = dr.regularization(Ae, Ai, Ar, o)
            (l + r1 + r2 + r3).backward()
        return run


def bench(device, steps=8, warmup=3, template=None, image_size=256, batch=48):
    """images/s of the trainer-shaped step at BASELINE config 3 (bench.py's value_config3)."""
    import os
    if template is None:
        template = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "templates", "ellipsoid.npz")
    ts = TrainerStep(template, image_size, batch, device)

    def timed(fn, n):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / n

    for _ in range(warmup):
        ts.step()
    t_step = timed(ts.step, steps)
    loss = float(ts.last["loss"])
    rp = ts.render_path_only()
    for _ in range(2):
        rp()
    t_rp = timed(rp, steps)
    # the same step with render #4 geometry-only (lean)
    tl = TrainerStep(template, image_size, batch, device, lean=True)
    for _ in range(warmup):
        tl.step()
    t_step_l = timed(tl.step, steps)
    rpl = tl.render_path_only()
    for _ in range(2):
        rpl()
    t_rp_l = timed(rpl, steps)
    nparam = sum(p.numel() for p in ts.netE.parameters())
    return {"workload": "config3: template ellipsoid (V=%d,F=%d), B=%d, %dx%d, texture %dx%d; ResNet-18 x2 + conv stacks (%.1f M params) -> "
                        "4 renders (trainer.py order) -> recon_data -> regularisers (chamfer IC) -> one backward -> Adam"
                        % (ts.dr.num_vertices, ts.dr.num_faces, batch, ts.dr.render_height, ts.dr.image_size, 2 * ts.dr.render_height,
                           ts.dr.image_size, nparam / 1e6),
            "images_per_s": round(batch / t_step, 1), "ms_per_step": round(t_step * 1e3, 3),
            "render_path_ms": round(t_rp * 1e3, 3), "render_path_share": round(t_rp / t_step, 3),
            "lean": {"images_per_s": round(batch / t_step_l, 1), "ms_per_step": round(t_step_l * 1e3, 3), "render_path_ms": round(t_rp_l * 1e3, 3),
                     "loss": float(tl.last["loss"]), "what": "render #4 (image discarded, trainer.py:367) as DiffRender.render_geometry"},
            "steps": steps, "loss": loss,
            "encoder_params": nparam}
