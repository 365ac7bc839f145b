"""CPU: host-side pieces around the hot path -- growth schedule, CLI flags, WAV IO, input transforms."""
import numpy as np
import pytest
import torch


def test_grower_schedule_matches_reference_rules():
    """utils.py:45-68 / SURVEY App. B #3: alpha = min(1, (1+step)/fadein[curr]); grow when cumsum(train)[curr] < samples
    (strict); after the 7th grow never again."""
    from musicgan_amd.utils import Grower
    fade = [1, 25000, 37500, 50000, 62500, 75000, 87500, 100000]
    train = [50000, 100000, 150000, 200000, 250000, 300000, 350000]
    g = Grower(n_grow=7, fadein_lengths=fade, train_lengths=train)
    assert g.alpha == 1.0
    cum = np.cumsum(train)
    seen, grows = 0, []
    bs = 6
    while len(grows) < 7:
        grew = g.grow(bs)
        seen += bs
        if grew:
            grows.append(seen)
            assert g.alpha == pytest.approx(1.0 / fade[len(grows)])
    for k, s in enumerate(grows):
        assert cum[k] < s <= cum[k] + bs
    assert not g.grow(10 ** 7)
    g2 = Grower(7, fade, train)
    g2.load_state_dict(g.state_dict())
    assert g2.alpha == g.alpha and g2.curr_grow == 7
    with pytest.raises(AssertionError):
        Grower(7, fade[:-1], train)


def test_grower_transform_range_and_size():
    from musicgan_amd.utils import Grower
    g = Grower(7, [1] * 8, [10] * 7)
    x = torch.rand(3, 2, 512, 512, dtype=torch.float64).float() * 5 - 2
    y = g.scale_transform(x)
    assert tuple(y.shape) == (3, 2, 4, 4)
    assert float(y.min()) >= -1.0 - 1e-5 and float(y.max()) <= 1.0 + 1e-5


def test_transforms_match_reference_formulas():
    from musicgan_amd.audio import ChangeRange, ChannelMinMaxNorm
    x = torch.randn(4, 2, 8, 8)
    y = ChannelMinMaxNorm()(x)
    flat = x.view(4, 2, -1)
    ref = (x - flat.min(-1)[0].view(4, 2, 1, 1)) / (flat.max(-1)[0].view(4, 2, 1, 1) - flat.min(-1)[0].view(4, 2, 1, 1) + 1e-8)
    assert torch.equal(y, ref)
    assert torch.equal(ChangeRange(-1., 1.)(y), y * 2. + -1.)
    with pytest.raises(AssertionError):
        ChannelMinMaxNorm()(torch.zeros(2, 3, 4, 4))


def test_transforms_match_reference_fixture():
    """tests/golden/transforms.npz: the reference's own ChannelMinMaxNorm / ChangeRange outputs (audio/transforms.py:4-40, run by
    tools/gen_golden.py) incl. a constant channel, and its Grower.scale_transform at three levels (Resize through the
    torchvision stand-in = aten bilinear + antialias)."""
    from golden_util import load
    from musicgan_amd.audio import ChangeRange, ChannelMinMaxNorm
    from musicgan_amd.utils import Grower
    g = load("transforms.npz")
    x = torch.from_numpy(g["x64"]).to(torch.float)
    norm = ChannelMinMaxNorm()(x)
    assert np.max(np.abs(norm.numpy() - g["norm"])) <= 1e-7
    assert np.max(np.abs(ChangeRange(-1.0, 1.0)(norm).numpy() - g["ranged"])) <= 2e-7
    assert np.all(g["norm"][1, 0] == 0.0)  # constant channel: (x - min) / (0 + eps) == 0
    big = torch.from_numpy(g["big32"]).double()
    grower = Grower(7, [1] * 8, [1] * 7)
    for level in range(6):
        if level in (0, 3, 5):
            y = grower.scale_transform(big.to(torch.float))
            assert np.max(np.abs(y.numpy() - g[f"scaled_l{level}"])) <= 2e-6, level
        grower.grow(2)


def test_cli_flags_are_the_reference_flags():
    from musicgan_amd.__main__ import build_parser
    p = build_parser()
    a = p.parse_args(["create_dataset", "/x/*.wav", "-o", "out"])
    assert (a.mode, a.audio_path, a.output_dir) == ("create_dataset", "/x/*.wav", "out")
    a = p.parse_args(["train", "run0", "-o", "out", "-i", "data"])
    assert (a.mode, a.run, a.out_path, a.input_dataset) == ("train", "run0", "out", "data")
    a = p.parse_args(["generate", "gen.pt", "32", "-o", "o"])
    assert (a.gen_dict_state, a.rand_channels, a.nb_vec, a.nb_music, a.output_dir) == ("gen.pt", 32, 10, 5, "o")
    a = p.parse_args(["view_audio", "--input-audio", "a.wav", "--image-idx", "3"])
    assert (a.input_audio, a.image_idx) == ("a.wav", 3)
    with pytest.raises(SystemExit):
        p.parse_args(["train", "run0"])


def test_wav_io_roundtrip(tmp_path):
    from musicgan_amd.audio import wavio
    x = (torch.rand(2, 4410) - 0.5)
    path = str(tmp_path / "a.wav")
    wavio.save(path, x, 44100)
    y, sr = wavio.load(path)
    assert sr == 44100 and torch.equal(x, y)
    from scipy.io import wavfile
    wavfile.write(str(tmp_path / "b.wav"), 22050, (x[0].numpy() * 32767).astype(np.int16))
    y, sr = wavio.load(str(tmp_path / "b.wav"))
    assert sr == 22050 and tuple(y.shape) == (1, 4410) and float((y[0] - x[0]).abs().max()) < 1e-4


def test_package_reexports_drivers_lazily():
    import musicgan_amd
    assert callable(musicgan_amd.train) and callable(musicgan_amd.generate)
    assert callable(musicgan_amd.create_dataset) and callable(musicgan_amd.view_audio)
    assert callable(musicgan_amd.train)  # still the function after the sub-module import
    from musicgan_amd.audio import N_FFT, N_VEC, SAMPLE_RATE, STFT_STRIDE
    assert (N_FFT, N_VEC, STFT_STRIDE, SAMPLE_RATE) == (1024, 512, 256, 44100)


def test_kernel_selection_predicates_respect_index_limits():
    """The engine only routes a layer to the Winograd / sub-pixel kernels when their 31/32-bit offset arithmetic can address it;
    the long non-square maps of `generate` (e.g. 512 x 5120*k) fall back to the direct kernels instead of failing."""
    from musicgan_amd import ops
    assert ops.wino3x3_supported(64, 64, 128, 128, cin=48)
    assert ops.wino3x3_supported(1, 48, 512, 5120, cin=48)
    assert not ops.wino3x3_supported(1, 48, 512, 5120 * 8, cin=64)       # 64 * 512 * 40960 * 4 B > 2^31
    assert not ops.wino3x3_supported(64, 80, 32, 32, pixnorm=True, cin=80)  # PixelNorm epilogue needs <= 64 channels
    assert not ops.wino3x3_supported(64, 64, 7, 8, cin=64) and not ops.wino3x3_supported(64, 64, 8, 8, ups=True, cin=64)
    assert not ops.wino3x3_supported(2, 64, 8, 8, cin=64)               # too few tiles
    assert ops.upconv3x3_supported(48, 64, 64 * 64 * 64 * 64) and not ops.upconv3x3_supported(48, 64, 1 << 31)
    assert not ops.upconv3x3_supported(96, 64, 1024)
    assert ops.wino_wgrad_supported(192, 48, 64, 128, 128) and not ops.wino_wgrad_supported(192, 48, 64, 127, 128)
    assert not ops.wino_wgrad_supported(1024, 64, 64, 128, 128)         # > 2^29 elements


def test_packed_sidecar_index_and_invalidation(tmp_path):
    """The float32 side-car of the fast loader (audio/dataset.py) without a GPU: rows may sit in ANY order in the array (create_dataset
    streams them out in write order, idx 0, 1, 2, ...) and are served in the reference's order -- `magn_phase_*.pt` names sorted as
    plain strings, so magn_phase_10.pt comes before magn_phase_2.pt (audio/dataset.py:14-44 of the reference); a side-car whose
    files were replaced (other sizes), whose array is truncated, or whose directory gained a file is not valid any more."""
    import json
    import os
    import importlib
    cd = importlib.import_module("musicgan_amd.create_dataset")  # (the package attribute of that name is the function)
    from musicgan_amd.audio import dataset as ds
    folder = str(tmp_path)
    shape = ds._SAMPLE_SHAPE
    rng = torch.Generator().manual_seed(1)
    n = 12
    samples = [(torch.rand(*shape, generator=rng) * 2 - 1) for _ in range(n)]
    for i, x in enumerate(samples):
        torch.save(x.double(), os.path.join(folder, f"magn_phase_{i}.pt"))
    # what create_dataset does: rows streamed in write order into the shards' .tmp files (blocks of PACKED_BLOCK_ROWS rows dealt
    # round-robin over PACKED_SHARDS files), then _finish_sidecar
    def stream(rows):
        fds = [open(os.path.join(folder, ds.shard_name(k, ds.PACKED_SHARDS) + ".tmp"), "wb") for k in range(ds.PACKED_SHARDS)]
        for r, x in enumerate(rows):
            k, local = ds.shard_of_row(r, ds.PACKED_SHARDS, ds.PACKED_BLOCK_ROWS)
            fds[k].seek(local * x.numel() * 4)
            fds[k].write(x.numpy().tobytes())
        for fh in fds:
            fh.close()
    stream(samples)
    cd._finish_sidecar(folder, [f"magn_phase_{i}.pt" for i in range(n)])
    assert ds.has_packed(folder)
    meta = json.load(open(os.path.join(folder, ds.PACKED_META)))
    assert meta["shards"] == ds.PACKED_SHARDS and os.path.getsize(os.path.join(folder, ds.shard_name(1, ds.PACKED_SHARDS))) == 4 * samples[0].numel() * 4
    packed, ref = ds.PackedAudioDataset(folder), ds.AudioDataset(folder)
    names = sorted(f"magn_phase_{i}.pt" for i in range(n))
    assert names[2] == "magn_phase_10.pt" and len(packed) == len(ref) == n
    for i in range(n):
        assert torch.equal(packed[i].double(), ref[i]), names[i]
    out = torch.empty((3,) + shape)
    packed.gather([2, 0, 11], out)
    assert torch.equal(out[0].double(), ref[2]) and torch.equal(out[2].double(), ref[11])
    # write_packed (rebuild from the .pt files) serves the same items
    assert ds.write_packed(folder) == n and ds.has_packed(folder)
    again = ds.PackedAudioDataset(folder)
    assert all(torch.equal(again[i], packed[i]) for i in range(n))
    meta = json.load(open(os.path.join(folder, ds.PACKED_META)))
    assert meta["rows"] == list(range(n)) and len(meta["sizes"]) == n
    # invalidation
    torch.save(torch.zeros(2, 2), os.path.join(folder, "magn_phase_3.pt"))            # replaced behind the side-car's back
    assert not ds.has_packed(folder)
    torch.save(samples[3].double(), os.path.join(folder, "magn_phase_3.pt"))
    assert ds.has_packed(folder)
    torch.save(samples[4].double(), os.path.join(folder, "magn_phase_3.pt"))          # same size, another sample
    assert not ds.has_packed(folder)
    torch.save(samples[3].double(), os.path.join(folder, "magn_phase_3.pt"))
    assert ds.has_packed(folder)
    with open(os.path.join(folder, ds.PACKED_BIN), "r+b") as fh:                       # truncated array
        fh.truncate(100)
    assert not ds.has_packed(folder)
    ds.write_packed(folder)
    torch.save(samples[0].double(), os.path.join(folder, "magn_phase_12.pt"))         # one more file
    assert not ds.has_packed(folder)
    # a run that wrote fewer files than the directory holds leaves no side-car (the loader then takes the reference path)
    stream(samples[:1])
    cd._remove_sidecar(folder)
    assert not [f for f in os.listdir(folder) if f.startswith(ds.PACKED_BIN)]
    stream(samples[:1])
    cd._finish_sidecar(folder, ["magn_phase_0.pt"])
    assert not ds.has_packed(folder) and not [f for f in os.listdir(folder) if f.endswith(".tmp")]


def test_pt_template_writes_what_torch_save_writes(tmp_path):
    """fast_pt.PtTemplate: `th.save(float64 tensor of a fixed shape)` from a template + payload + the payload's CRC-32 (which
    create_dataset takes from the GPU), byte for byte -- incl. the zip data descriptor, the central directory and the
    serialization_id record torch derives from the records' CRCs -- for the sample shape (create_dataset.py:52-62) and others."""
    import io
    import zlib
    from musicgan_amd.fast_pt import PtTemplate
    rng = torch.Generator().manual_seed(3)
    for shape in ((2, 512, 512), (2, 512, 64), (5, 3)):
        tpl = PtTemplate(shape)
        assert tpl.ok, shape
        for _ in range(2):
            x = (torch.rand(*shape, generator=rng) * 2 - 1).double()
            want = io.BytesIO()
            torch.save(x, want)
            path = str(tmp_path / "s.pt")
            tpl.write(path, x.numpy(), zlib.crc32(x.numpy().tobytes()) & 0xFFFFFFFF)
            assert open(path, "rb").read() == want.getvalue()
            assert torch.equal(torch.load(path), x)


def test_sleef_restatement_has_the_bits_of_torch_cpu_abs_and_angle(tmp_path):
    """csrc/sleef_f32.h (what the codec kernel evaluates th.abs / th.angle with, functions.py:69-70) compiled for the host by g++
    and compared BITWISE with torch's CPU `abs` / `angle` on complex64 -- the library the reference itself calls -- over 2 M
    log-uniform random bins plus zeros, signed zeros, denormals, huge values, infinities and NaNs.  (The device build of the same
    header is pinned by tests/test_audio_gpu.py against the reference's golden codec output.)"""
    import ctypes
    import os
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "musicgan_amd", "csrc")
    shim = tmp_path / "shim.cpp"
    shim.write_text('#define SLF_FN static inline\n#include "sleef_f32.h"\n'
                    'extern "C" void slf_angle_n(const float* re, const float* im, float* o, long n) '
                    '{ for (long i = 0; i < n; ++i) o[i] = slf::atan2f_u10(im[i], re[i]); }\n'
                    'extern "C" void slf_abs_n(const float* re, const float* im, float* o, long n) '
                    '{ for (long i = 0; i < n; ++i) o[i] = slf::hypotf_u05(re[i], im[i]); }\n'
                    'extern "C" void slf_both_n(const float* re, const float* im, float* m, float* a, long n) '
                    '{ for (long i = 0; i < n; ++i) slf::abs_angle(re[i], im[i], m[i], a[i]); }\n')
    so = str(tmp_path / "shim.so")
    flags = ["-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I", csrc]
    with open("/proc/cpuinfo") as f:
        if " fma " in f.read():
            flags.append("-mfma")  # (without it __builtin_fmaf calls libm's correctly rounded fmaf: same bits, slower)
    subprocess.run(["g++", *flags, str(shim), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(2)
    n = 1 << 21
    re = (rng.standard_normal(n) * np.exp(rng.uniform(-12, 6, n))).astype(np.float32)
    im = (rng.standard_normal(n) * np.exp(rng.uniform(-12, 6, n))).astype(np.float32)
    sp = np.array([0, -0.0, 1, -1, np.inf, -np.inf, 1e-40, -1e-40, 3e38, -3e38, 1e-30, np.nan, 1e30, 1e-31, .5, -.5], np.float32)
    re[:256], im[:256] = np.repeat(sp, 16), np.tile(sp, 16)
    re[256:1256] = im[256:1256]                      # |re| == |im|
    re[1256:9000] *= np.float32(1e-28)               # around the combined routine's range gate and into the denormals
    im[5000:9000] *= np.float32(1e-28)
    re[9000:12000] *= np.float32(1e27)
    c = torch.complex(torch.from_numpy(re), torch.from_numpy(im))
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)  # one chunk whose length is a multiple of the vector width: ATen's vector routine on every element
    try:                      # (the remainder elements of a thread's chunk go through libm instead: oracle/audio.py _abs_angle)
        wants = (torch.angle(c).numpy(), torch.abs(c).numpy())
    finally:
        torch.set_num_threads(nthr)
    for fn, want in zip((lib.slf_angle_n, lib.slf_abs_n), wants):
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long]
        out = np.empty(n, np.float32)
        fn(re.ctypes.data, im.ctypes.data, out.ctypes.data, n)
        same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
        assert bool(same.all()), [(re[i], im[i], out[i], want[i]) for i in np.nonzero(~same)[0][:5]]
    # the codec kernel's entry: both at once, sharing the quotient (same bits)
    m, a = np.empty(n, np.float32), np.empty(n, np.float32)
    lib.slf_both_n.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long]
    lib.slf_both_n(re.ctypes.data, im.ctypes.data, m.ctypes.data, a.ctypes.data, n)
    for out, want in ((a, wants[0]), (m, wants[1])):
        same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
        assert bool(same.all()), [(re[i], im[i], out[i], want[i]) for i in np.nonzero(~same)[0][:5]]


def test_aiff_and_au_files_load_like_wav_files(tmp_path):
    """`th_audio.load` (functions.py:43) reads whatever torchaudio can; without torchaudio the uncompressed containers are read
    here: AIFF and Sun AU (big-endian linear PCM, 8 / 16 / 24 / 32 bits) through the standard library -- same normalisation as
    a WAV file of the same samples (torchaudio.load(normalize=True): v / 2^(bits-1)), `load_pcm` keeps them integer for the
    device path, anything codec-backed raises."""
    import aifc
    import sunau
    from musicgan_amd.audio import wavio
    rng = np.random.default_rng(5)
    frames, ch = 1000, 2
    for width in (1, 2, 3, 4):
        bits = 8 * width
        v = rng.integers(-(1 << (bits - 1)), (1 << (bits - 1)) - 1, (frames, ch), dtype=np.int64)
        raw = b"".join(int(s).to_bytes(width, "big", signed=True) for s in v.reshape(-1))
        want = (v.astype(np.float64) / float(1 << (bits - 1))).astype(np.float32).T
        for mod, ext in ((aifc, ".aiff"), (sunau, ".au")):
            path = str(tmp_path / f"t{bits}{ext}")
            with mod.open(path, "wb") as f:
                f.setnchannels(ch)
                f.setsampwidth(width)
                f.setframerate(44100)
                if mod is sunau:
                    f.setcomptype("NONE", "not compressed")  # (the module's default is u-law)
                f.writeframes(raw)
            x, sr = wavio.load(path)
            assert sr == 44100 and tuple(x.shape) == (ch, frames) and x.dtype == torch.float32
            assert np.array_equal(x.numpy(), want), (bits, ext)
            pcm, sr = wavio.load_pcm(path)
            assert pcm.shape == (frames, ch) and pcm.dtype == (np.int16 if width <= 2 else np.int32)
    with pytest.raises(ValueError, match="flac"):
        wavio.load(str(tmp_path / "song.flac"))


def test_bench_with_gpus_n_and_no_launcher_starts_the_ranks_as_a_child_process(tmp_path):
    """`python bench.py --gpus 8` the way the driver runs N = 1 (no torchrun, no WORLD_SIZE): bench.py must start
    `python -m torch.distributed.run --nproc-per-node 8 ... bench.py <same flags>` itself, as a child, BEFORE it touches the GPU
    (this container has none: reaching `torch.cuda.is_available()` would exit with "needs an MI355X"), hand its stdout through and
    return its exit code.  MG_BENCH_LAUNCHER swaps the launcher for a recording stub."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = tmp_path / "stub_launcher.py"
    stub.write_text("import json, os, sys\n"
                    f"json.dump({{'argv': sys.argv[1:], 'ipc': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}}, open({str(tmp_path / 'seen.json')!r}, 'w'))\n"
                    "print(json.dumps({'metric': 'from the stub', 'n_gpus': 8}))\n"
                    "sys.exit(7)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MG_BENCH_LAUNCHER"] = f"{sys.executable} {stub}"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode == 7, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"metric": "from the stub", "n_gpus": 8}
    seen = json.load(open(tmp_path / "seen.json"))
    argv = seen["argv"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert "--standalone" in argv and argv[argv.index("--local-addr") + 1] == "127.0.0.1"  # (the launcher binds its own port)
    i = argv.index(os.path.join(root, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert seen["ipc"] == "0"
    # under a launcher (RANK set) it does not spawn again: it goes on to the GPU check, which fails here -- and a rank that a driver
    # started with `torch.distributed.run ... bench.py --gpus 8` directly (never through launch_ranks) has set the IPC mode itself
    env2 = {k: v for k, v in dict(env, RANK="0", WORLD_SIZE="8", LOCAL_RANK="0").items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env2,
                        cwd=root, timeout=300)
    assert r2.returncode != 0 and "needs an MI355X" in (r2.stderr + r2.stdout)
    assert "HSA_ENABLE_IPC_MODE_LEGACY=0" in (r2.stderr + r2.stdout)


def test_native_sample_writer_writes_what_torch_save_writes(tmp_path):
    """mg_pt_write_samples (the writer threads' native call, host code only): n float32 rows -> n `.pt` files byte-identical to
    `th.save(row.to(th.float64))` (reference create_dataset.py:52-62) + the float32 side-car rows at their offsets."""
    import ctypes
    import io
    import os
    import zlib
    from musicgan_amd import _lib
    from musicgan_amd.fast_pt import PtTemplate
    shape = (2, 16, 8)
    tp = PtTemplate(shape)
    assert tp.ok
    g = torch.Generator().manual_seed(3)
    rows = torch.randn(5, *shape, generator=g)
    rows[1, 0, 0, 0], rows[2, 1, 3, 3] = -0.0, 1e-42  # signed zero, a denormal
    paths = [str(tmp_path / f"s_{i}.pt") for i in range(5)]
    crcs = [zlib.crc32(r.double().numpy().tobytes()) & 0xFFFFFFFF for r in rows]
    suffixes = b"".join(tp.suffix(c) for c in crcs)
    side = os.open(str(tmp_path / "side.bin"), os.O_RDWR | os.O_CREAT, 0o644)
    off0 = 3 * rows[0].numel() * 4  # the batch starts at side-car row 3
    rc = _lib.load().mg_pt_write_samples(ctypes.c_void_p(rows.data_ptr()), 5, rows[0].numel(),
                                         b"\0".join(os.fsencode(p) for p in paths) + b"\0", tp.prefix, len(tp.prefix), suffixes,
                                         len(suffixes) // 5, side, off0)
    assert rc == 0
    for i, p in enumerate(paths):
        ref = io.BytesIO()  # (th.save names the archive's root after its target: a stream gives the template's "archive")
        torch.save(rows[i].to(torch.float64), ref)
        assert open(p, "rb").read() == ref.getvalue(), i
        assert torch.equal(torch.load(p), rows[i].double())
    got = np.fromfile(str(tmp_path / "side.bin"), dtype=np.float32)
    assert got.size == 8 * rows[0].numel() and np.array_equal(got[3 * rows[0].numel():].view(np.uint32),
                                                              rows.numpy().reshape(-1).view(np.uint32))
    os.close(side)
    # an unwritable path is an error, not a crash
    rc = _lib.load().mg_pt_write_samples(ctypes.c_void_p(rows.data_ptr()), 1, rows[0].numel(), os.fsencode(str(tmp_path / "no" / "x.pt")) + b"\0",
                                         tp.prefix, len(tp.prefix), suffixes, len(suffixes) // 5, -1, 0)
    assert rc != 0 and b"cannot open" in _lib.load().mg_last_error()
