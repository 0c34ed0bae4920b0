"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
exactly the entry points include/deephumor_hip.h declares, with matching arity (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _prototypes():
    text = open(os.path.join(ROOT, "include", "deephumor_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|double|const char\*)\s+(dh_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return protos


def test_header_declares_the_path():
    protos = _prototypes()
    for name in ("dh_conv2d_bn_act", "dh_maxpool3x3s2", "dh_linear", "dh_embed_rows", "dh_add_layernorm",
                 "dh_attn_self_decode", "dh_attn_cross_decode", "dh_lstm_cell", "dh_beam_row_sample",
                 "dh_beam_select", "dh_beam_finalize"):
        assert name in protos


def test_library_builds_loads_and_exports_every_symbol():
    from deephumor_amd import _build, hip
    path = _build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    protos = _prototypes()
    for name, nargs in protos.items():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        if name in hip.SIGNATURES:
            assert len(hip.SIGNATURES[name]) == nargs, f"{name}: binding has {len(hip.SIGNATURES[name])} args, header {nargs}"
    for name in hip.SIGNATURES:
        assert name in protos, f"{name} bound but not declared in include/deephumor_hip.h"
    lib.dh_abi_version.restype = ctypes.c_int
    assert lib.dh_abi_version() == hip.ABI_VERSION
    lib.dh_error_string.restype = ctypes.c_char_p
    assert lib.dh_error_string(1).startswith(b"bad argument")


def test_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation happens before any HIP call, so it is testable on CPU."""
    from deephumor_amd import hip
    lib = hip.load()
    assert lib.dh_linear(None, 0, None, 0, None, None, None, None, 0, None, 0, 4, 4, 4, 0, hip.F32, None) == 1
    assert lib.dh_maxpool3x3s2(None, None, 1, 1, 4, 4, 7, None) == 2
    assert lib.dh_conv2d_bn_act(None, None, None, None, None, None, 1, 3, 8, 8, 8, 5, 5, 1, 0, 1, hip.F32, None) == 1
    assert lib.dh_beam_row_sample(None, 0, 10, 1, 1, 3, 2, 1.0, 1, None, 0, None, 0, 0, None, None, None, None) == 1


def test_every_entry_point_rejects_all_zero_arguments():
    """Argument contracts are checked before anything is launched: every compute entry point called with NULL pointers and zero
    sizes returns an error code (no GPU needed, nothing is dereferenced)."""
    import ctypes
    from deephumor_amd import hip
    lib = hip.load()
    called = 0
    for name, sig in hip.SIGNATURES.items():
        if (name in ("dh_abi_version", "dh_option_count") or name.endswith("_supported") or name.endswith("_bytes") or name.startswith("dh_prof")
                or name == "dh_strerror"):
            continue
        args = []
        for t in sig:
            if t in (ctypes.c_float, ctypes.c_double):
                args.append(0.0)
            elif isinstance(getattr(t, "_type_", None), type) or t is ctypes.c_char_p:            # POINTER(struct), const char*
                args.append(None)
            else:
                args.append(0)
        assert getattr(lib, name)(*args) != 0, name
        called += 1
    assert called >= 55


def test_option_table():
    """dh_set_option / dh_get_option: one table for every kernel-selection switch; defaults from the environment at first use, a set
    value is read back, an unknown key is an argument error, and changing an option re-keys the models' weight plans."""
    from deephumor_amd import hip
    opts = hip.options()
    assert 10 <= len(opts) <= 20 and all(env == "DH_" + key.upper() for key, (_, env) in opts.items())
    for key in ("decode_wreg", "decode_wreg_min_rows", "lstm_wreg", "vocab_wreg", "f32_split", "encoder_generic", "deferred_ln", "dist_always"):
        assert key in opts, key
    lib = hip.load()
    assert lib.dh_set_option(b"no_such_option", 1) == 1 and lib.dh_get_option(b"no_such_option", None) == 1
    epoch = hip.options_epoch
    old = hip.set_option("decode_wreg_min_rows", 123)
    assert hip.option("decode_wreg_min_rows") == 123 and hip.options_epoch == epoch + 1
    with hip.option_scope(decode_wreg_min_rows=7, encoder_generic=2):
        assert hip.option("decode_wreg_min_rows") == 7 and hip.option("encoder_generic") == 2
    assert hip.option("decode_wreg_min_rows") == 123 and hip.option("encoder_generic") == opts["encoder_generic"][0]
    hip.set_option("decode_wreg_min_rows", old)
    assert hip.option("decode_wreg_min_rows") == opts["decode_wreg_min_rows"][0]
