"""`check_inputs` error behaviour of the diffusers-style pipeline (utils/stable_diffusion_controlnet_inpaint.py:792-979):
same exception types on the same conditions.  Host-only (the checks run before anything touches the device)."""
import types

import numpy as np
import pytest
import torch
from PIL import Image

from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline as Pipe


def make(n_controlnets=None, in_channels=4):
    p = object.__new__(Pipe)
    nets = [object()] * (n_controlnets or 1)
    p.controlnet = nets if n_controlnets else nets[0]
    p.controlnets = nets
    p.unet = types.SimpleNamespace(cfg={"in_channels": in_channels})
    return p


PE = torch.zeros(1, 77, 8)
IMG = Image.fromarray(np.zeros((64, 64, 3), np.uint8))
MSK = Image.fromarray(np.zeros((64, 64), np.uint8))


def call(p, **over):
    kw = dict(prompt=None, image=IMG, mask_image=MSK, cond_images=IMG, height=64, width=64, callback_steps=1,
              negative_prompt=None, prompt_embeds=PE, negative_prompt_embeds=PE, cond_scale=1.0)
    kw.update(over)
    return p.check_inputs(**kw)


def test_valid_inputs_pass():
    call(make())
    call(make(2), cond_images=[IMG, IMG], cond_scale=[1.0, 0.5])
    call(make(2), cond_images=[IMG, IMG], cond_scale=0.7)            # a scalar is broadcast over the nets
    call(make(), image=None, mask_image=None)                        # generation pipeline: no inpaint inputs
    t_img, t_msk = torch.zeros(2, 3, 64, 64), torch.ones(2, 1, 64, 64)
    call(make(in_channels=9), image=t_img, mask_image=t_msk)


@pytest.mark.parametrize("over,exc", [
    (dict(height=60), ValueError),                                   # :806-809
    (dict(callback_steps=0), ValueError),                            # :811-818
    (dict(callback_steps=None), ValueError),
    (dict(callback_steps=1.5), ValueError),
    (dict(prompt="a cat"), ValueError),                              # both prompt and prompt_embeds :820-824
    (dict(prompt_embeds=None, negative_prompt_embeds=None), ValueError),   # neither :825-828
    (dict(prompt=3, prompt_embeds=None, negative_prompt_embeds=None), ValueError),   # wrong prompt type :829-834
    (dict(negative_prompt="bad"), ValueError),                       # both negatives :836-840
    (dict(negative_prompt_embeds=torch.zeros(1, 60, 8)), ValueError),   # shape mismatch :842-848
    (dict(cond_scale=[1.0]), TypeError),                             # single net: scale must be float :875-879
    (dict(cond_scale=1), TypeError),
    (dict(image=torch.zeros(3, 64, 64)), TypeError),                 # tensor image with a PIL mask :891-894
    (dict(mask_image=torch.zeros(64, 64)), TypeError),               # PIL image with a tensor mask :896-901
    (dict(image=None), ValueError),
])
def test_single_controlnet_errors(over, exc):
    with pytest.raises(exc):
        call(make(), **over)


@pytest.mark.parametrize("over,exc", [
    (dict(cond_images=IMG), TypeError),                              # must be a list :857-859
    (dict(cond_images=(IMG, IMG)), TypeError),
    (dict(cond_images=[IMG]), ValueError),                           # wrong length :860-863
    (dict(cond_images=[IMG, IMG], cond_scale=[1.0]), ValueError),    # scale list of the wrong length :880-887
])
def test_multi_controlnet_errors(over, exc):
    with pytest.raises(exc):
        call(make(2), **over)


@pytest.mark.parametrize("image,mask,msg", [
    (torch.zeros(64, 64), torch.zeros(64, 64), "3 or 4 dimensions"),
    (torch.zeros(3, 64, 64), torch.zeros(1, 1, 1, 64, 64), "2, 3, or 4 dimensions"),
    (torch.zeros(1, 64, 64), torch.zeros(64, 64), "3 channels"),
    (torch.zeros(2, 3, 64, 64), torch.zeros(2, 2, 64, 64), "1 channel"),
    (torch.zeros(2, 3, 64, 64), torch.zeros(3, 64, 64), "batch sizes"),
    (torch.zeros(3, 64, 64), torch.zeros(32, 64), "height and width"),
    (torch.full((3, 64, 64), 1.5), torch.zeros(64, 64), r"range \[-1, 1\]"),
    (torch.zeros(3, 64, 64), torch.full((64, 64), 2.0), r"range \[0, 1\]"),
])
def test_tensor_image_and_mask_checks(image, mask, msg):
    with pytest.raises(ValueError, match=msg):
        call(make(), image=image, mask_image=mask)


def test_unet_channel_check():
    with pytest.raises(ValueError, match="expects 5"):
        call(make(in_channels=5))


def test_call_front_half_with_bench_shaped_inputs():
    """bench.py's call (tensor image [4,3,512,512] in [-1,1], one mask / control image / prompt row per image, CFG) through
    everything `__call__` does on the host before the first device op -- validation, prompt / control / latent /
    inpaint-input preparation -- with the device parts stubbed out."""
    from editanything_amd.scheduler import DDIMScheduler

    class Stop(Exception):
        pass

    seen = {}
    p = make()
    p.unet.plan = {"input": [[("conv_in", 4, 8)]] * 12}

    def encode(img, noise):
        seen["vae_in"], seen["vae_noise"] = tuple(img.shape), tuple(noise.shape)
        return torch.zeros(img.shape[0], 4, 64, 64)

    def compute_invariants(embeds, hints):
        seen["embeds"], seen["hints"] = tuple(embeds.shape), [tuple(h.shape) for h in hints]
        return {}

    def time_embeddings(ts):
        seen["timesteps"] = len(ts)
        return []

    def install(inv, per_net, static=None):
        seen["scales"] = per_net
        raise Stop()

    p.vae = types.SimpleNamespace(encode=encode, scale_factor=0.18215)
    p.scheduler, p.text_encoder, p.tokenizer, p.device = DDIMScheduler(), None, None, torch.device("cpu")
    p.denoiser = types.SimpleNamespace(compute_invariants=compute_invariants, time_embeddings=time_embeddings, install=install,
                                       only_mid_control=False)
    p.use_graph, p._graphs, p.trace = True, {}, None
    mask = torch.zeros(1, 1, 512, 512)
    mask[:, :, 128:384, 128:384] = 1
    with pytest.raises(Stop):
        p(prompt_embeds=torch.zeros(4, 77, 1024), negative_prompt_embeds=torch.zeros(4, 77, 1024),
          image=torch.rand(4, 3, 512, 512) * 2 - 1, mask_image=mask.repeat(4, 1, 1, 1),
          controlnet_conditioning_image=torch.zeros(4, 3, 512, 512), height=512, width=512, num_inference_steps=20,
          guidance_scale=7.5, num_images_per_prompt=1, generator=torch.Generator("cpu").manual_seed(0))
    assert seen["vae_in"] == (4, 3, 512, 512) and seen["vae_noise"] == (4, 4, 64, 64)
    assert seen["embeds"] == (8, 77, 1024) and seen["hints"] == [(8, 3, 512, 512)]          # [uncond || cond]
    assert seen["scales"] == [[1.0] * 13] and seen["timesteps"] == 20


def test_load_textual_inversion_adds_tokens_and_rows(tmp_path):
    """`pipe.load_textual_inversion(file)` (reference call editany_lora.py:733-735; diffusers TextualInversionLoaderMixin):
    both file layouts, multi-vector embeddings -> token, token_1, ...; the rows land in the text encoder's table."""
    import torch
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline

    class Tok:
        def __init__(self):
            self.v = {"a": 0, "b": 1, "c": 2}

        def get_vocab(self):
            return dict(self.v)

        def add_tokens(self, toks):
            for t in toks:
                self.v[t] = len(self.v)

        def convert_tokens_to_ids(self, toks):
            return [self.v[t] for t in toks]

        def __len__(self):
            return len(self.v)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(3, 8)

        def get_input_embeddings(self):
            return self.emb

        def resize_token_embeddings(self, n):
            new = torch.nn.Embedding(n, 8)
            new.weight.data[:self.emb.num_embeddings] = self.emb.weight.data
            self.emb = new

    pipe = StableDiffusionControlNetInpaintPipeline.__new__(StableDiffusionControlNetInpaintPipeline)
    pipe.tokenizer, pipe.text_encoder = Tok(), Enc()
    old = pipe.text_encoder.emb.weight.data.clone()
    v1 = torch.randn(2, 8)
    torch.save({"string_to_param": {"*": v1}, "name": "<cat>"}, tmp_path / "a1111.pt")
    assert pipe.load_textual_inversion(str(tmp_path / "a1111.pt")) == ["<cat>", "<cat>_1"]
    w = pipe.text_encoder.get_input_embeddings().weight.data
    assert torch.equal(w[:3], old) and torch.equal(w[3:5], v1) and len(pipe.tokenizer) == 5
    v2 = torch.randn(8)
    torch.save({"<dog>": v2}, tmp_path / "plain.bin")
    assert pipe.load_textual_inversion(str(tmp_path / "plain.bin")) == ["<dog>"]
    assert torch.equal(pipe.text_encoder.get_input_embeddings().weight.data[5], v2)
    with pytest.raises(ValueError):
        pipe.load_textual_inversion(str(tmp_path / "plain.bin"))          # token already there
    pipe.tokenizer = None
    with pytest.raises(ValueError):
        pipe.load_textual_inversion(str(tmp_path / "plain.bin"))


# ---------------------------------------------------------------------------- serving.merge_kwargs (host-only part)
def _merge_req(seed, b=2, eta=0.0, img_rows=None, gen_list=False):
    g = torch.Generator("cpu").manual_seed(1000 + seed)
    rows = b if img_rows is None else img_rows
    gens = [torch.Generator("cpu").manual_seed(10 * seed + i) for i in range(b)] if gen_list else torch.Generator("cpu").manual_seed(seed)
    return dict(prompt_embeds=torch.randn(b, 77, 8, generator=g), negative_prompt_embeds=torch.randn(b, 77, 8, generator=g),
                image=torch.rand(rows, 3, 64, 64, generator=g) * 2 - 1, mask_image=(torch.rand(rows, 1, 64, 64, generator=g) > 0.5).float(),
                controlnet_conditioning_image=torch.rand(b, 3, 64, 64, generator=g), height=64, width=64, num_inference_steps=4,
                guidance_scale=7.5, eta=eta, output_type="latent", generator=gens)


def test_merge_kwargs_concatenates_requests_and_keeps_each_request_its_own_draws():
    """serving.predraw + merge_kwargs: N requests -> one batched call whose rows are the requests' rows in order, whose `latents` /
    `vae_noise` are exactly what each request's own call would draw from ITS generator (x_T first, then the VAE posterior noise:
    …inpaint.py:1005-1007, 1079-1081; a list of generators = one per image for x_T, the first for the VAE noise) -- taken the
    moment the request's kwargs exist, so a generator object the NEXT request re-seeds (torch.manual_seed: sam2image.py:163-167)
    cannot disturb them; num_images_per_prompt > 1 expands prompt-major; a group that cannot be merged returns None."""
    from editanything_amd import serving
    from editanything_amd.pipeline import randn_tensor
    pipe = types.SimpleNamespace(unet=types.SimpleNamespace(cfg={"in_channels": 4}), device=torch.device("cpu"), text_encoder=None)
    reqs = [_merge_req(1), _merge_req(2, img_rows=1), _merge_req(3, gen_list=True)]
    drawn = [serving.predraw(pipe, kw) for kw in reqs]
    merged, sizes = serving.merge_kwargs(pipe, drawn)
    assert sizes == [2, 2, 2] and merged["generator"] is None and merged["latents"].shape == (6, 4, 8, 8) and merged["num_images_per_prompt"] == 1
    fresh = [_merge_req(1), _merge_req(2, img_rows=1), _merge_req(3, gen_list=True)]
    lo = 0
    for r, kw in enumerate(fresh):
        g = kw["generator"]
        if isinstance(g, list):
            lat = torch.cat([randn_tensor((1, 4, 8, 8), gi, "cpu") for gi in g])
            vn = randn_tensor((kw["image"].shape[0], 4, 8, 8), g[0], "cpu")
        else:
            lat = randn_tensor((2, 4, 8, 8), g, "cpu")
            vn = randn_tensor((kw["image"].shape[0], 4, 8, 8), g, "cpu")
        assert torch.equal(drawn[r]["latents"], lat) and torch.equal(drawn[r]["vae_noise"], vn), r     # own-call form: own shapes
        assert torch.equal(merged["latents"][lo:lo + 2], lat), r
        assert torch.equal(merged["vae_noise"][lo:lo + 2], vn.expand(2, -1, -1, -1)), r
        assert torch.equal(merged["prompt_embeds"][lo:lo + 2], kw["prompt_embeds"])
        assert torch.equal(merged["image"][lo:lo + 2], kw["image"].expand(2, -1, -1, -1))
        assert torch.equal(merged["controlnet_conditioning_image"][lo:lo + 2], kw["controlnet_conditioning_image"])
        lo += 2
    # the reference's seeding: every request re-seeds the GLOBAL generator; draws taken per request are the sequential ones
    kws = []
    for seed in (5, 6):
        kw = dict(_merge_req(seed), generator=torch.manual_seed(seed))
        kws.append(serving.predraw(pipe, kw))
    for seed, d in zip((5, 6), kws):
        assert torch.equal(d["latents"], randn_tensor((2, 4, 8, 8), torch.manual_seed(seed), "cpu"))
    # num_images_per_prompt = 3, one prompt / one control / one image per request (sam2image.process's form): prompt-major rows
    nip = [dict(_merge_req(s, b=1), num_images_per_prompt=3) for s in (7, 8)]
    m2, sz2 = serving.merge_kwargs(pipe, [serving.predraw(pipe, kw) for kw in nip])
    assert sz2 == [3, 3] and m2["prompt_embeds"].shape[0] == 6 and m2["image"].shape[0] == 6 and m2["latents"].shape[0] == 6
    assert torch.equal(m2["prompt_embeds"][3:], nip[1]["prompt_embeds"].expand(3, -1, -1))
    # string prompts need the pipeline's text encoder
    txt = [dict(_merge_req(s), prompt=["a", "b"], prompt_embeds=None, negative_prompt_embeds=None) for s in (1, 2)]
    assert serving.merge_kwargs(pipe, [serving.predraw(pipe, kw) for kw in txt]) is None
    enc = types.SimpleNamespace(unet=pipe.unet, device=pipe.device, text_encoder=object(),
                                _encode_text=lambda ps: torch.stack([torch.full((77, 8), float(len(p_))) for p_ in ps]))
    m3, _ = serving.merge_kwargs(enc, [serving.predraw(enc, kw) for kw in txt])
    assert m3["prompt_embeds"].shape == (4, 77, 8) and float(m3["negative_prompt_embeds"].abs().max()) == 0.0   # "" -> length 0
    # unmergeable alone: per-call state (reference-only control, callbacks, scale maps) -> no predraw at all, generator untouched
    for kw in (dict(_merge_req(2), ref_image=object()), dict(_merge_req(2), callback=print), dict(_merge_req(2), controlnet_conditioning_scale_map=torch.zeros(1))):
        assert not serving.mergeable_alone(kw)
    assert serving.mergeable_alone(_merge_req(1)) and serving.mergeable_alone(_merge_req(2, eta=0.3)) and serving.mergeable_alone(dict(_merge_req(2), alpha_weight=0.5))
    # eta > 0 / mixing: the LOOP's draws are taken too, right after x_T and the VAE noise, and handed over as `loop_noise=`
    lp = types.SimpleNamespace(unet=pipe.unet, device=pipe.device, text_encoder=None, loop_draws=lambda steps, eta, alpha, has_image: 7)
    de = [serving.predraw(lp, _merge_req(s_, eta=0.3)) for s_ in (11, 12)]
    g = _merge_req(11, eta=0.3)["generator"]
    want = [randn_tensor((2, 4, 8, 8), g, "cpu") for _ in range(9)]           # x_T, VAE noise, 7 loop draws: one stream, this order
    assert torch.equal(de[0]["latents"], want[0]) and torch.equal(de[0]["vae_noise"], want[1]) and len(de[0]["loop_noise"]) == 7
    assert all(torch.equal(a, b) for a, b in zip(de[0]["loop_noise"], want[2:]))
    me, sze = serving.merge_kwargs(lp, de)
    assert sze == [2, 2] and len(me["loop_noise"]) == 7 and me["loop_noise"][3].shape == (4, 4, 8, 8) and me["eta"] == 0.3
    assert torch.equal(me["loop_noise"][3][:2], de[0]["loop_noise"][3]) and torch.equal(me["loop_noise"][3][2:], de[1]["loop_noise"][3])
    assert serving.predraw(pipe, _merge_req(3, eta=0.3)) is None             # (a pipeline that cannot say how many: the call draws for itself)
    assert serving.merge_kwargs(lp, [serving.predraw(lp, _merge_req(1, eta=0.3)), serving.predraw(lp, _merge_req(2))]) is None     # eta differs
    for bad in ([_merge_req(1), dict(_merge_req(2), height=128, width=128)], [_merge_req(1), dict(_merge_req(2), num_inference_steps=5)],
                [_merge_req(1), _merge_req(2, b=3, img_rows=2)], [_merge_req(1)]):
        assert serving.merge_kwargs(pipe, [serving.predraw(pipe, kw) for kw in bad]) is None
    assert serving.merge_kwargs(types.SimpleNamespace(unet=types.SimpleNamespace(cfg={"in_channels": 9}), device=torch.device("cpu")),
                                [serving.predraw(pipe, kw) for kw in (_merge_req(1), _merge_req(2))]) is None
    out = serving.split_output(types.SimpleNamespace(images=torch.arange(6)), [2, 1, 3])
    assert [o.images.tolist() for o in out] == [[0, 1], [2], [3, 4, 5]]


def test_runner_regroups_a_queue_by_compatibility_class():
    """serving.PipelinedRunner(merge=2, regroup=True): the mergeable requests of a queue are gathered by compatibility class
    (`merge_key`: size, steps, scales, inpaint or not ...) before units are formed -- [A512, B768, A512, B768, A512] runs as
    (A, A), (A), (B, B) -- and every request's output comes back at ITS place.  A generator object shared by two requests (or
    `generator=None`: the global one) pins the order: no regrouping then."""
    from editanything_amd import serving

    class FakePipe:
        device = torch.device("cpu")
        unet = types.SimpleNamespace(cfg={"in_channels": 4})
        text_encoder = None

        def __init__(self):
            self.fronts = []

        def front(self, **kw):
            self.fronts.append((kw["height"], kw["prompt_embeds"].shape[0]))
            if kw.get("latents") is None:           # a call of its own draws for itself, like the pipeline
                kw = dict(kw, latents=torch.randn((1, 4, kw["height"] // 8, kw["width"] // 8), generator=kw["generator"]))
            return types.SimpleNamespace(kw=kw, gkey=None)

        def has_graph(self, c):
            return True

        def loop(self, c):
            c.final = c.kw["latents"][:, :1, :1, :1].flatten() + c.kw["height"]        # one number per image: its first latent + the size

        def back(self, c):
            return types.SimpleNamespace(images=c.final)
    mk = lambda res, seed: dict(prompt_embeds=torch.zeros(1, 77, 8), negative_prompt_embeds=torch.zeros(1, 77, 8), height=res, width=res,
                                controlnet_conditioning_image=torch.zeros(1, 3, res, res), num_inference_steps=4, guidance_scale=7.5,
                                generator=torch.Generator().manual_seed(seed))
    sizes = [512, 768, 512, 768, 512]
    want = [float(torch.randn((1, 4, s // 8, s // 8), generator=torch.Generator().manual_seed(i))[0, 0, 0, 0]) + s for i, s in enumerate(sizes)]
    pipe = FakePipe()
    outs = serving.PipelinedRunner(pipe, merge=2, regroup=True).run([mk(s, i) for i, s in enumerate(sizes)])
    assert pipe.fronts == [(512, 2), (512, 1), (768, 2)], pipe.fronts
    assert np.allclose([float(o.images[0]) for o in outs], want, atol=1e-3)
    pipe2 = FakePipe()
    outs2 = serving.PipelinedRunner(pipe2, merge=2, regroup=False).run([mk(s, i) for i, s in enumerate(sizes)])
    assert pipe2.fronts == [(512, 1), (768, 1), (512, 1), (768, 1), (512, 1)]          # consecutive pairs never match: own calls
    assert np.allclose([float(o.images[0]) for o in outs2], want, atol=1e-3)
    shared = torch.Generator().manual_seed(0)
    pipe3 = FakePipe()
    serving.PipelinedRunner(pipe3, merge=2, regroup=True).run([dict(mk(s, i), generator=shared) for i, s in enumerate(sizes)])
    assert [f[0] for f in pipe3.fronts] == sizes, "a shared generator pins the order of execution"
    assert serving.merge_key(pipe, mk(512, 0)) == serving.merge_key(pipe, mk(512, 9)) != serving.merge_key(pipe, mk(768, 0))
    assert serving.merge_key(pipe, dict(mk(512, 0), callback=print)) is None
