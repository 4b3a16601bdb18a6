"""Layer-wise learning-rate decay groups of finetune.py:378-383 (reference lr_decay.py:15-76, the BEiT rule): a parameter of layer
`i` of `L = depth + 1` trains at `layer_decay ** (L - i)` of the base rate; tokens and the patch embedding are layer 0, block `b` is
layer `b + 1`, everything after the blocks layer L.  Host-side grouping only; the scale reaches the device through each group's lr."""


def get_layer_id_for_vit(name, num_layers):
    if name in ('cls_token', 'pos_embed') or name.startswith('patch_embed'):
        return 0
    if name.startswith('blocks'):
        return int(name.split('.')[1]) + 1
    return num_layers


def param_groups_lrd(model, weight_decay=0.05, no_weight_decay_list=[], layer_decay=.75):
    num_layers = len(model.blocks) + 1
    scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    groups = {}
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        no_decay = p.ndim == 1 or n in no_weight_decay_list          # 1-D tensors and the model's own list
        layer = get_layer_id_for_vit(n, num_layers)
        key = 'layer_%d_%s' % (layer, 'no_decay' if no_decay else 'decay')
        if key not in groups:
            groups[key] = {'lr_scale': scales[layer], 'weight_decay': 0. if no_decay else weight_decay, 'params': []}
        groups[key]['params'].append(p)
    return list(groups.values())
