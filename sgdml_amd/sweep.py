"""
Hyper-parameter sweep driver: the train -> validate -> select -> test loop the reference runs from its CLI
(`sgdml all`: sgdml/cli.py:993-1148 training loop with early stopping, :1514-1640 validation / test error
accumulation, :1797-1872 selection by the lowest validation force RMSE), as a library function over the
accelerated path.  BASELINE.json configs[0] (`sgdml all ethanol_dft.npz 200 1000 5000`) is this loop.

What is shared across the sigma grid: the sampled task, the descriptors and Jacobians,
the device-resident training set with its dense tables and permutation tables (uploaded once, content-hashed
in _lib.Context.train_upload), the kernel-matrix buffer (reused when the size is unchanged) and the validation
geometries.  Validation and test errors are reduced on the device (gdml_predict_errors); per model 64 bytes
come back.
"""
import timeit

import numpy as np

from .predict import GDMLPredict


def _errors_into_model(model, errs, n_pts, is_test, md5=None):
    """Write the error fields like cli.py:1686-1712 does."""
    model['f_err'] = {'mae': float(errs['force'][0]), 'rmse': float(errs['force'][1])}
    if model['use_E'] and 'energy' in errs:
        model['e_err'] = {'mae': float(errs['energy'][0]), 'rmse': float(errs['energy'][1])}
    if is_test:
        model['n_test'] = int(n_pts)
        model['md5_test'] = md5


def sigma_sweep(gdml_train, dataset, n_train, n_valid, n_test, sigs=None, valid_dataset=None, test_dataset=None,
                lam=1e-10, perms=None, use_sym=True, use_E=True, use_E_cstr=False, early_stop=True, callback=None,
                emulate_cli_rng=False):
    """Train one model per sigma on a shared task, validate each, select the best, test it.

    Returns (best_model, table, timings): `table` rows are (sig, e_mae, e_rmse, f_mae, f_rmse) of the validated
    models in training order (the columns `sgdml select` prints, cli.py:1852-1854); `timings` holds wall-clock
    seconds of the train / validate / test phases.

    The train / validation / test samples come from the global NumPy stream like the reference's.  `emulate_cli_rng`
    additionally consumes what a freshly installed reference CLI draws between them for its CPU worker benchmark (one
    rand(min(1000, n_valid), 3N) per validation until its cache holds three runs: cli.py:1513-1518 -> predict.py:833-858,
    :1087-1090), so that a seed reproduces the reference's test sample too (tests/test_hip_r3.py).
    """
    if sigs is None:
        sigs = list(range(10, 100, 10))  # cli.py:806: default grid '10:10:100'
    valid_dataset = dataset if valid_dataset is None else valid_dataset
    test_dataset = dataset if test_dataset is None else test_dataset

    t_task = timeit.default_timer()
    task0 = gdml_train.create_task(dataset, n_train, valid_dataset, n_valid, sigs[0], lam=lam, perms=perms,
                                   use_sym=use_sym, use_E=use_E, use_E_cstr=use_E_cstr, callback=callback)
    t_task = timeit.default_timer() - t_task
    n_atoms = task0['R_train'].shape[1]
    iv = task0['idxs_valid']
    R_valid = np.ascontiguousarray(valid_dataset['R'][iv].reshape(len(iv), -1))
    F_valid = np.ascontiguousarray(valid_dataset['F'][iv].reshape(len(iv), -1))
    E_valid = valid_dataset['E'][iv] if (use_E and 'E' in valid_dataset) else None

    models, table = [], []
    t_train = t_valid = 0.0
    prev_err, converged_once = -1.0, False
    for sig in sigs:
        task = dict(task0, sig=sig)
        t0 = timeit.default_timer()
        model = gdml_train.train(task, callback=callback)
        t1 = timeit.default_timer()
        pred = GDMLPredict(model, _borrow_ctx=gdml_train._context())  # short-lived: the trainer's context serves it
        errs = pred.test_errors(R_valid, F_valid, E_valid)
        del pred
        # the reference shuffles the validation indices of every model before its online error loop (cli.py:1487-1488);
        # the device reduction does not care about the order, but the draw keeps the global NumPy stream -- which picks
        # the test sample below -- where the reference's is
        np.random.shuffle(np.array(iv))
        if emulate_cli_rng and len(models) < 3:
            np.random.rand(min(1000, len(iv)), 3 * n_atoms)
        t2 = timeit.default_timer()
        t_train += t1 - t0
        t_valid += t2 - t1
        _errors_into_model(model, errs, len(iv), is_test=False)
        models.append(model)
        e = errs.get('energy', (0.0, 0.0))
        table.append((sig, float(e[0]), float(e[1]), float(errs['force'][0]), float(errs['force'][1])))
        is_conv = True
        if 'solver_resid' in model:
            is_conv = model['solver_resid'] <= model['solver_tol'] * model['norm_y_train']
        converged_once = converged_once or is_conv
        # cli.py:1136-1147: stop once the validation error rises again; valid_errs there is what cli.test returns, the
        # list of force RMSEs of the models it was given (cli.py:1792-1794), so valid_errs[0] is this model's force RMSE
        lead = float(errs['force'][1])
        if early_stop and converged_once and prev_err != -1.0 and prev_err < lead:
            break
        prev_err = lead

    f_rmse = [row[4] for row in table]
    best = models[f_rmse.index(min(f_rmse))]  # cli.py:1871-1872

    t3 = timeit.default_timer()
    n_tested = 0
    if n_test != 0:
        excl = np.empty((0,), dtype=np.uint)  # cli.py:1441-1451
        from .utils import io

        md5_test = io.dataset_md5(test_dataset)
        if md5_test == best['md5_train']:
            excl = np.concatenate([excl, best['idxs_train']]).astype(np.uint)
        if md5_test == best['md5_valid']:
            excl = np.concatenate([excl, best['idxs_valid']]).astype(np.uint)
        n_eff = test_dataset['F'].shape[0] - len(excl)
        n_tested = n_eff if n_test < 0 else min(n_test, n_eff)
        if n_tested > 0:
            if 'E' in test_dataset:
                it = gdml_train.draw_strat_sample(test_dataset['E'], n_tested, excl_idxs=excl)
            else:
                it = np.delete(np.arange(test_dataset['F'].shape[0]), excl)[:n_tested]
            pred = GDMLPredict(best, _borrow_ctx=gdml_train._context())
            errs = pred.test_errors(test_dataset['R'][it].reshape(len(it), -1), test_dataset['F'][it].reshape(len(it), -1),
                                    test_dataset['E'][it] if (use_E and 'E' in test_dataset) else None)
            del pred
            _errors_into_model(best, errs, len(it), is_test=True, md5=md5_test)
    t_test = timeit.default_timer() - t3
    timings = {'create_task_s': t_task, 'train_s': t_train, 'validate_s': t_valid, 'test_s': t_test,
               'n_models': len(models), 'n_valid': int(len(iv)), 'n_test': int(n_tested), 'n_atoms': int(n_atoms)}
    return best, table, timings
