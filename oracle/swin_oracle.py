"""TEST INFRASTRUCTURE (imported by tests/ only).  The reference's image tower is HF `SwinModel` (reference swin.py:119-149 calls
`SwinModel.from_pretrained('microsoft/swin-tiny-patch4-window7-224')(..., output_hidden_states=True)`); transformers is a third-party
dependency of the reference that is installed in this image, so the oracle for the Swin-T tower IS that model on the CPU in fp32:
random weights of the published geometry (the checkpoint cannot be fetched offline), eval mode (stochastic depth off)."""
import torch


def make_swin(seed: int = 0, spread: bool = True):
    """SwinModel(SwinConfig()) = swin-tiny-patch4-window7-224 geometry.  spread: move LayerNorm parameters, biases and the relative position
    tables away from their 1 / 0 / tiny initial values so that every term of the tower carries signal; GEMM weights rounded to bf16
    (the HIP tower keeps bf16 working copies of them)."""
    from transformers import SwinConfig, SwinModel
    torch.manual_seed(seed)
    model = SwinModel(SwinConfig()).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if spread:
                if "relative_position_bias_table" in name:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.5)
                elif "norm" in name and name.endswith(".weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                elif name.endswith(".bias"):
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
                elif p.dim() >= 2:
                    p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p[0].numel()) ** 0.5)       # unit-gain Linear / conv
            if p.dim() >= 2 and "relative_position_bias_table" not in name:
                p.copy_(p.to(torch.bfloat16).float())
    return model


def swin_forward(model, images: torch.Tensor):
    """-> (hidden_states[0..3], last_hidden_state, pooled) exactly as reference swin.py:136-143 reads them."""
    out = model(pixel_values=images, output_hidden_states=True)
    return [out.hidden_states[i] for i in range(4)], out.last_hidden_state, out.pooler_output


def set_drop_path_masks(model, masks):
    """Train-mode stochastic depth with GIVEN keep masks (one [B] tensor of 0 / 1 per block, None = block not dropped), in place of
    SwinDropPath.forward's torch.rand draw (modeling_swin.py): hidden_states / keep_prob * mask."""
    import torch.nn as nn

    class Fixed(nn.Module):
        def __init__(self, mask, keep):
            super().__init__()
            self.mask, self.keep = mask, keep

        def forward(self, x):
            return x / self.keep * self.mask.view(-1, *([1] * (x.dim() - 1)))

    rates = torch.linspace(0, model.config.drop_path_rate, sum(model.config.depths)).tolist()
    i = 0
    for stage in model.encoder.layers:
        for blk in stage.blocks:
            blk.drop_path = nn.Identity() if masks[i] is None else Fixed(masks[i].float(), 1.0 - rates[i])
            i += 1
    return model
