# Search space of the R50..R101(+) seg supernet and its train / val samplers.
# Same space as the reference's configs/_dynamic_/model_samplers/ar50to101v2.py:2-116
# (stem 32..64 step 16; stage widths 48-80 / 96-160 / 192-320 / 384-640; depths 2-4 / 2-6 / 5-29 /
# 2-4; anchors MAX, MIN, R101, R77, R50; train = 5 anchors + 3 random draws per step).
_stem = dict(key='arch.backbone.stem.width', start=32, end=64, step=16)
_width = dict(key='arch.backbone.body.width', start=[48, 96, 192, 384], end=[80, 160, 320, 640],
              step=[16, 32, 64, 128], ascending=True)
_depth = dict(key='arch.backbone.body.depth', start=[2, 2, 5, 2], end=[4, 6, 29, 4],
              step=[1, 2, 2, 1])


def _anchor(name, stem, width, depth):
    return {'name': name, 'arch.backbone.stem.width': stem, 'arch.backbone.body.width': width,
            'arch.backbone.body.depth': depth}


_MAX = _anchor('MAX', _stem['end'], _width['end'], _depth['end'])
_MIN = _anchor('MIN', _stem['start'], _width['start'], _depth['start'])
_R50 = _anchor('R50', 64, [64, 128, 256, 512], [3, 4, 6, 3])
_R77 = _anchor('R77', 64, [64, 128, 256, 512], [3, 4, 15, 3])
_R101 = _anchor('R101', 64, [64, 128, 256, 512], [3, 4, 23, 3])

train_sampler = dict(
    type='concat',
    model_samplers=[
        dict(type='anchor', anchors=[_MAX, _MIN, _R101, _R77, _R50]),
        dict(type='repeat', times=3, model_sampler=dict(
            type='composite',
            model_samplers=[dict(type='range', **_stem), dict(type='range', **_width),
                            dict(type='range', **_depth)])),
    ])
val_sampler = dict(type='anchor', anchors=[_R50, _R77, _R101])
