"""Golden vectors from the reference's OWN third-party stack, for the semantics the oracle restates without being able to run
them (tensorflow==2.2.0, tensorflow-addons==0.10.0, environment.yml:14-16).  Neither wheel can be installed in the
build image (no network), so this script is committed UNRUN: on any machine that has both,

    python tests/golden/make_tf_golden.py            # writes tests/golden/tf_ops.npz (~60 KB) and tests/golden/tf_ckpt/ (~20 KB)

and tests/test_oracle_third_party.py::test_oracle_ops_against_tensorflow_golden then checks oracle/tf_ops.py and
oracle/nlt_oracle.py against TensorFlow's outputs (it is skipped while the file is absent).  Covered: Conv2D / Conv2DTranspose
'same' for kernel 1 / 2 and stride 1 / 2 (nlt/networks/elements.py:26-39), LeakyReLU(0.3) (:69-78), tfa.image.resampler incl.
the (-1, 0) and (W-1, W) border bands, exact integers and fp16-rounded coordinates (nlt/models/nlt.py:112-114), tf.image.resize
(nlt/util/img.py:115), three Adam(amsgrad=True) steps (nlt/trainvali.py:122-127) and what `clipnorm` does when the loop calls
tape.gradient + apply_gradients (:279-280); and a REAL tensor-bundle checkpoint written by `tf.train.Checkpoint(net=...)` over the
attribute names the reference's models register (`net_{query,obs}_layer{i}`, nlt/models/base.py:79-101; nlt/trainvali.py:134-141),
for tests/test_host_ckpt.py::test_reader_against_a_checkpoint_written_by_tensorflow -- the reader (ckpt.py) has so far only met
bundles written by the test suite's own writer."""
import os

import numpy as np


def main():
    import tensorflow as tf
    import tensorflow_addons as tfa
    assert tf.__version__.startswith('2.2'), tf.__version__
    rng = np.random.default_rng(0)
    out = {}
    f32 = lambda a: np.asarray(a, np.float32)

    # ---- Conv2D / Conv2DTranspose 'same'
    for tr in (False, True):
        for k, s in ((1, 1), (2, 1), (2, 2)):
            if tr and k == 1:
                continue
            cin, cout, h, w = 5, 4, 6, 8
            x = f32(rng.standard_normal((2, h, w, cin)))
            layer = (tf.keras.layers.Conv2DTranspose if tr else tf.keras.layers.Conv2D)(cout, k, strides=s, padding='same')
            layer.build(x.shape)
            wk = f32(rng.standard_normal(layer.kernel.shape) * 0.5); b = f32(rng.standard_normal(cout))
            layer.set_weights([wk, b])
            name = 'conv_%s_k%d_s%d' % ('t' if tr else 'f', k, s)
            name = ('conv_t' if tr else 'conv_f') + '_k%d_s%d' % (k, s)
            out[name + '_x'], out[name + '_w'], out[name + '_b'], out[name + '_y'] = x, wk, b, layer(x).numpy()

    # ---- LeakyReLU
    x = f32(rng.standard_normal((3, 7)))
    out['lrelu_x'], out['lrelu_y'] = x, tf.keras.layers.LeakyReLU(alpha=0.3)(x).numpy()

    # ---- tfa.image.resampler
    h, w = 9, 13
    data = f32(rng.standard_normal((2, h, w, 3)))
    xy = rng.uniform(-2.0, 1.0, (2, 20, 11, 2)) + rng.uniform(0, 1, (2, 20, 11, 2)) * [w + 1.0, h + 1.0]
    xy[0, :3, :3] = np.round(xy[0, :3, :3])
    xy[0, 4, :, 0] = -0.5; xy[0, 5, :, 0] = w - 0.5; xy[0, 6, :, 1] = -0.25; xy[0, 7, :, 1] = h - 0.75
    xy[0, 8, 0] = (-1.0, 3.0); xy[0, 8, 1] = (float(w), 3.0); xy[0, 8, 2] = (w - 1.0, h - 1.0)
    xy[1] = xy[1].astype(np.float16)                          # save_float16_npy coordinates (data_gen/render.py:155)
    warp = f32(xy)
    out['resampler_data'], out['resampler_warp'] = data, warp
    out['resampler_out'] = tfa.image.resampler(data, warp).numpy()

    # ---- tf.image.resize
    x = f32(rng.standard_normal((2, 8, 12, 3)))
    out['resize_x'], out['resize_y'] = x, tf.image.resize(x, (5, 19)).numpy()

    # ---- Adam(amsgrad=True), three steps with given gradients
    p0 = f32(rng.standard_normal((4, 3)))
    v = tf.Variable(p0)
    opt = tf.keras.optimizers.Adam(learning_rate=1e-3, amsgrad=True)
    out['adam_p0'], out['adam_lr'] = p0, np.float32(1e-3)
    for i in range(3):
        g = f32(rng.standard_normal((4, 3)) * (10.0 ** (-i)))
        opt.apply_gradients([(tf.constant(g), v)])
        out['adam_g%d' % i], out['adam_p%d' % (i + 1)] = g, v.numpy()

    # ---- clipnorm through tape.gradient + apply_gradients (what nlt/trainvali.py does)
    g = f32(rng.standard_normal((4, 3)) * 5.0)
    norm = np.float32(1.0)
    out['clip_g'], out['clip_norm'] = g, norm
    out['clip_by_norm_out'] = tf.clip_by_norm(tf.constant(g), norm).numpy()
    res = {}
    for label, grad in (('applied', g), ('if_clipped', out['clip_by_norm_out'])):
        v = tf.Variable(p0)
        o = tf.keras.optimizers.SGD(learning_rate=0.1, clipnorm=float(norm) if label == 'applied' else None)
        o.apply_gradients([(tf.constant(grad), v)])
        res[label] = v.numpy()
    out['clip_p1_applied'], out['clip_p1_if_clipped'] = res['applied'], res['if_clipped']
    # DESIGN.md section 8 claims apply_gradients does NOT clip in TF 2.2 (clipping sits in get_gradients / minimize):
    clipped = bool(np.allclose(res['applied'], res['if_clipped'], rtol=1e-6, atol=1e-9))
    out['clip_apply_gradients_clips'] = np.array(clipped)
    if clipped:
        print("NOTE: apply_gradients DID clip here -- DESIGN.md section 8's claim is wrong for this TF build; fix "
              "`clip_apply_gradients_clips` and make clipping the default (mgm_apply = true).")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tf_ops.npz')
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(out), "arrays; tensorflow", tf.__version__, "tensorflow_addons", tfa.__version__)

    # ---- a real checkpoint: plain Conv2D layers and Keras Sequential blocks under the reference's attribute names
    class Net(tf.keras.Model):
        pass
    net = Net()
    expect = {}
    conv = lambda cout, k, s: tf.keras.layers.Conv2D(cout, k, strides=s, padding='same')
    deconv = lambda cout, k, s: tf.keras.layers.Conv2DTranspose(cout, k, strides=s, padding='same')
    specs = {'query': [('plain', conv(4, 1, 1), 5), ('seq', [conv(6, 2, 2), conv(6, 2, 1)], 4), ('seq', [deconv(4, 2, 2), deconv(4, 2, 1)], 6),
                       ('plain', conv(3, 1, 1), 8)],
             'obs': [('plain', conv(4, 1, 1), 3), ('seq', [conv(6, 2, 2), conv(6, 2, 1)], 4)]}
    for path_name, layers in specs.items():
        for i, (kind, lay, cin) in enumerate(layers):
            attr = 'net_%s_layer%d' % (path_name, i)
            if kind == 'plain':
                lay.build((None, 8, 8, cin))
                lay.set_weights([f32(rng.standard_normal(w_.shape)) for w_ in lay.get_weights()])
                setattr(net, attr, lay)
                expect['net/%s/kernel/.ATTRIBUTES/VARIABLE_VALUE' % attr] = lay.get_weights()[0]
                expect['net/%s/bias/.ATTRIBUTES/VARIABLE_VALUE' % attr] = lay.get_weights()[1]
            else:
                seq = tf.keras.Sequential([lay[0], tf.keras.layers.LeakyReLU(), lay[1], tf.keras.layers.LeakyReLU()])
                seq.build((None, 8, 8, cin))
                for j, l_ in enumerate(lay):
                    l_.set_weights([f32(rng.standard_normal(w_.shape)) for w_ in l_.get_weights()])
                    expect['net/%s/layer_with_weights-%d/kernel/.ATTRIBUTES/VARIABLE_VALUE' % (attr, j)] = l_.get_weights()[0]
                    expect['net/%s/layer_with_weights-%d/bias/.ATTRIBUTES/VARIABLE_VALUE' % (attr, j)] = l_.get_weights()[1]
                setattr(net, attr, seq)
    ckdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tf_ckpt')
    os.makedirs(ckdir, exist_ok=True)
    prefix = tf.train.Checkpoint(step=tf.Variable(7), net=net).save(os.path.join(ckdir, 'ckpt'))
    np.savez_compressed(os.path.join(ckdir, 'expected.npz'), prefix=np.array(os.path.basename(prefix)),
                        **{k.replace('/', '|'): v for k, v in expect.items()})
    print("wrote", prefix, "(+ expected.npz) with", len(expect), "variables")


if __name__ == '__main__':
    main()
