"""ASE calculator on top of the MI355X prediction path (reference: `sgdml/intf/ase_calc.py:25-106`).

Same constructor arguments, unit handling and `results` dictionary as the reference's
`SGDMLCalculator`; the model stays resident on the GPU and every `calculate` is one
single-geometry `GDMLPredict.predict` call (~70 us host-to-host for a 21-atom, 1000-point model,
`profiles/r01_latency_probe.txt`).  ASE is an optional dependency exactly as in the reference.
"""
import logging

import numpy as np

try:
    from ase.calculators.calculator import Calculator
    from ase.units import kcal, mol
except ImportError:  # same message and behaviour as the reference (ase_calc.py:25-31)
    raise ImportError("Optional ASE dependency not found! Please run 'pip install sgdml[ase]' to install it.")

from ..predict import GDMLPredict


class SGDMLCalculator(Calculator):
    implemented_properties = ['energy', 'forces']

    def __init__(self, model_path, E_to_eV=kcal / mol, F_to_eV_Ang=kcal / mol, use_torch=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.log = logging.getLogger(__name__)
        model = np.load(model_path, allow_pickle=True)
        self.gdml_predict = GDMLPredict(model, use_torch=use_torch)
        self.gdml_predict.prepare_parallel(n_bulk=1)
        self.log.warning(
            "Please remember to specify the proper conversion factors, if your model does not use 'kcal/mol' and "
            "'Ang' as units."
        )
        self.E_to_eV = E_to_eV                  # model energy unit -> eV
        self.Ang_to_R = F_to_eV_Ang / E_to_eV   # Angstrom -> model length unit
        self.F_to_eV_Ang = F_to_eV_Ang          # model force unit -> eV/Ang

    def calculate(self, atoms=None, *args, **kwargs):
        super().calculate(atoms, *args, **kwargs)
        r = np.array(atoms.get_positions()) * self.Ang_to_R
        e, f = self.gdml_predict.predict(r.ravel())
        self.results = {'energy': e * self.E_to_eV, 'forces': (f * self.F_to_eV_Ang).reshape(-1, 3)}
