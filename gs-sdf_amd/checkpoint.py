"""`local_map_checkpoint.pt`: the reference's SDF-network checkpoint (NeuralSLAM::export_checkpoint / load_checkpoint,
/root/reference/include/neural_mapping/neural_mapping.cpp:1331-1378): `torch::save(local_map_ptr, path)` of the LocalMap
module, i.e. a libtorch serialize archive (a TorchScript zip) whose entries are the module's registered parameters:

    "encoder_local_map"   flat fp32 hash-grid table          (register_parameter, local_map.cpp:73-75; name encoding_map.cpp:25)
    "decoder"             decoder_implementation 1: flat fp32 FullyFusedMLP weights (register_parameter, local_map.cpp:53-54)
                          decoder_implementation 0: submodule torch::nn::Sequential "decoder" with children "0".."2L+2"
                          (Linear / ReLU alternating, local_map.cpp:29-42) holding "weight" [out,in] and "bias" [out]

The same archive is written here with torch.jit (what libtorch's OutputArchive produces and InputArchive::load_from reads),
so `torch::load(local_map_ptr, path)` of a reference build LINKED AGAINST THE DROP-IN `tcnn_binding` accepts it and files such a
build wrote load here; tests/test_checkpoint_pt.py round-trips both directions through a libtorch C++ program, and tests/test_reference_intree_pins.py through
the REFERENCE'S OWN LocalMap module (compiled from /root/reference by oracle/ref_link, linked with the drop-in tcnn_binding).

Flat "decoder" layouts (decoder_implementation 1).  The drop-in TCNNNetwork keeps the weights UNPADDED, [out, in] row-major layer
after layer (last layer 2 x 64).  Upstream tiny-cuda-nn's FullyFusedMLP pads the output width to a multiple of 16 (last layer
16 x 64; rows >= n_out are dead): a checkpoint of that size is accepted on load (the real rows are taken) and
`save_local_map_checkpoint(..., pad_tcnn_output=True)` writes it.  Not checked against an upstream build (none is available here):
its parameter dtype (fp16 when tiny-cuda-nn is built with half precision) and the hash table's layout.
The rest of a checkpoint directory — gs.ply (neural_gs.export_gs_to_ply) and as_occ_prior.ply (LocalMap.export_as_occ_prior) —
already exists."""
import torch

TCNN_OUT_PAD = 16      # tiny-cuda-nn FullyFusedMLP: padded_output_width = next_multiple(n_output_dims, 16)


def _layers(lm):
    """[(weight [out,in], bias [out] or None), ...] of the decoder, whatever its implementation."""
    dec = lm.decoder
    if isinstance(dec, torch.nn.Module):
        return [(m.weight.detach(), m.bias.detach()) for m in dec if isinstance(m, torch.nn.Linear)]
    out, wo, bo = [], 0, 0
    for i, o in zip(dec.dims[:-1], dec.dims[1:]):
        w = dec.params_.detach()[wo:wo + i * o].view(o, i)
        b = None if dec.biases_ is None else dec.biases_.detach()[bo:bo + o]
        out.append((w, b))
        wo, bo = wo + i * o, bo + o
    return out


def save_local_map_checkpoint(lm, path, pad_tcnn_output=False):
    """Writes what `torch::save(local_map_ptr, path)` writes for this LocalMap (see the module docstring); pad_tcnn_output: the
    flat decoder parameter in upstream tiny-cuda-nn's layout (last layer padded to 16 output rows with zeros)."""
    root = torch.nn.Module()
    root.register_parameter("encoder_local_map", torch.nn.Parameter(lm.encoder.params_.detach().reshape(-1).cpu().clone()))
    layers = _layers(lm)
    if lm.decoder_implementation == 1:
        if any(b is not None for _, b in layers):
            raise RuntimeError("decoder_implementation 1 is bias free")
        ws = [w.cpu() for w, _ in layers]
        if pad_tcnn_output and ws[-1].shape[0] % TCNN_OUT_PAD:
            pad = TCNN_OUT_PAD - ws[-1].shape[0] % TCNN_OUT_PAD
            ws[-1] = torch.cat([ws[-1], torch.zeros(pad, ws[-1].shape[1])], 0)
        root.register_parameter("decoder", torch.nn.Parameter(torch.cat([w.reshape(-1) for w in ws]).clone()))
    else:
        mods = []
        for k, (w, b) in enumerate(layers):
            lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=True)
            with torch.no_grad():
                lin.weight.copy_(w.cpu())
                lin.bias.copy_(torch.zeros(w.shape[0]) if b is None else b.cpu())
            mods.append(lin)
            if k + 1 < len(layers):
                mods.append(torch.nn.ReLU(True))
        root.add_module("decoder", torch.nn.Sequential(*mods))
    torch.jit.script(root).save(str(path))


def load_local_map_checkpoint(lm, path):
    """`torch::load(local_map_ptr, path)`: copies the archive's parameters into this LocalMap (shapes must match)."""
    m = torch.jit.load(str(path), map_location="cpu")
    params = dict(m.named_parameters())
    if "encoder_local_map" not in params:
        raise RuntimeError(f"{path}: no 'encoder_local_map' parameter (not a LocalMap checkpoint)")
    enc = params["encoder_local_map"]
    if enc.numel() != lm.encoder.params_.numel():
        raise RuntimeError(f"{path}: hash-grid table has {enc.numel()} entries, this map {lm.encoder.params_.numel()}")
    layers = _layers(lm)
    # everything is validated BEFORE anything is copied: a refused checkpoint leaves the map untouched
    copies = [(lm.encoder.params_, enc.reshape(lm.encoder.params_.shape))]
    if "decoder" in params:                                  # decoder_implementation 1: one flat parameter
        flat = params["decoder"].reshape(-1).float()
        n_plain = sum(w.numel() for w, _ in layers)
        last = layers[-1][0]
        rows_padded = -(-last.shape[0] // TCNN_OUT_PAD) * TCNN_OUT_PAD
        n_padded = n_plain - last.numel() + rows_padded * last.shape[1]
        if any(b is not None for _, b in layers) or flat.numel() not in (n_plain, n_padded):
            raise RuntimeError(f"{path}: flat decoder parameter ({flat.numel()} values) does not fit this map's decoder "
                               f"({n_plain} unpadded / {n_padded} in tiny-cuda-nn's padded layout"
                               + (", and this map's decoder has biases: decoder_implementation 0 expects the Sequential layout)" if any(b is not None for _, b in layers) else ")"))
        padded = flat.numel() == n_padded and n_padded != n_plain
        off = 0
        for k, (w, _) in enumerate(layers):
            if padded and k + 1 == len(layers):               # upstream layout: [rows_padded, in], the real rows first
                copies.append((w, flat[off:off + rows_padded * w.shape[1]].view(rows_padded, w.shape[1])[:w.shape[0]]))
                off += rows_padded * w.shape[1]
            else:
                copies.append((w, flat[off:off + w.numel()].view_as(w)))
                off += w.numel()
    else:                                                    # Sequential: decoder.<2k>.weight / .bias
        for k, (w, b) in enumerate(layers):
            src_w, src_b = params.get(f"decoder.{2 * k}.weight"), params.get(f"decoder.{2 * k}.bias")
            if src_w is None or tuple(src_w.shape) != tuple(w.shape):
                raise RuntimeError(f"{path}: decoder.{2 * k}.weight missing or of the wrong shape")
            copies.append((w, src_w))
            if b is not None:
                if src_b is None or src_b.numel() != b.numel():
                    raise RuntimeError(f"{path}: decoder.{2 * k}.bias missing or of the wrong shape")
                copies.append((b, src_b))
            elif src_b is not None and float(src_b.detach().abs().max()) != 0.0:
                raise RuntimeError(f"{path}: the checkpoint's decoder has biases, this map's decoder is bias free")
    with torch.no_grad():
        for dst, src in copies:
            dst.copy_(src.to(dst.device))
    return lm
