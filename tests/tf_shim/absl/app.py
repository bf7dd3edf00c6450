def run(main):
    raise RuntimeError("tests/tf_shim: the reference's command-line drivers are not run, only their functions")
