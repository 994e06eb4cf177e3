"""Driver run in a SUBPROCESS by tests/test_reference_live.py (only where /root/reference exists): constructs the
reference's own model class over the TF-1.3 op shim and either

    restore <kind> <data_dir> <checkpoint.pickle> <out.npz>
        builds the reference model with --restore <checkpoint> (chem_tensorflow.py:330-359: by-name assignment of every
        global variable, the params assertion of :336-340) and dumps all its global variables to <out.npz>;
    train <kind> <data_dir> <out_dir>
        runs the reference's train() for the configured epochs and leaves its best-model pickle + log in <out_dir>.

A separate process keeps the shim's `tensorflow` module out of the test process."""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    mode, kind, data_dir = sys.argv[1:4]
    sys.path[:0] = [os.path.join(ROOT, "oracle", "tf13_shim"), "/root/reference"]
    import numpy as np
    import tensorflow as tf
    assert tf.__version__.endswith("shim")
    import chem_tensorflow
    chem_tensorflow.json = types.SimpleNamespace(                    # numpy-2 float32 scalars in the log (see make_reference_golden.py)
        dump=lambda o, f, **kw: json.dump(o, f, default=float, **kw), dumps=json.dumps, load=json.load, loads=json.loads)
    from chem_tensorflow_dense import DenseGGNNChemModel
    from chem_tensorflow_sparse import SparseGGNNChemModel
    cls = {"sparse": SparseGGNNChemModel, "dense": DenseGGNNChemModel}[kind]
    with open(os.path.join(data_dir, "config.json")) as f:
        config = f.read()
    if mode == "restore":
        ckpt, out = sys.argv[4:6]
        model = cls({"--data_dir": data_dir, "--log_dir": data_dir, "--config": config, "--restore": ckpt})
        names = [v.name for v in model.sess.graph.get_collection(tf.GraphKeys.GLOBAL_VARIABLES)]
        values = model.sess.run(model.sess.graph.get_collection(tf.GraphKeys.GLOBAL_VARIABLES))
        np.savez(out, names=np.array(names), train_step=model.train_step_id, valid_step=model.valid_step_id,
                 **{"v%d" % i: np.asarray(a) for i, a in enumerate(values)})
    else:
        out_dir = sys.argv[4]
        model = cls({"--data_dir": data_dir, "--log_dir": out_dir, "--config": config})
        model.train()
        with open(os.path.join(out_dir, "paths.json"), "w") as f:
            json.dump({"best": model.best_model_file, "log": model.log_file}, f)


if __name__ == "__main__":
    main()
