"""Body of __graft_entry__.smoke(): one tiny pass of the hot path on cuda:0, checked against the oracle.
Lives under tests/ because it uses the oracle (which only tests / smoke / bench's cpu_baseline may do)."""


def run(torch, np):
    import synth
    from melspec_gpt_vqvae_amd.vqvae.quantizer import vq_lookup
    from oracle import vq_c

    dev = "cuda:0"
    z = synth.normal(5, (2, 256, 5, 53))
    E = synth.normal(6, (128, 256))
    r = vq_lookup(torch.from_numpy(z).to(dev), torch.from_numpy(E).to(dev))
    ref = vq_c.vq_argmin_f32(np.ascontiguousarray(z.transpose(0, 2, 3, 1).reshape(-1, 256)), E)
    torch.cuda.synchronize()
    assert np.array_equal(r["indices"].cpu().numpy(), ref["indices"]), "VQ argmin mismatch vs oracle"
    print("smoke: vq argmin (2 tiles) bit-exact vs oracle")
    try:
        import smoke_gpt
    except ImportError:
        return
    smoke_gpt.run(torch, np)
