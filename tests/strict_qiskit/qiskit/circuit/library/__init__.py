"""TEST INFRASTRUCTURE -- qiskit.circuit.library of the strict double: the standard gates the path
meets and ``AND`` (QCMRF.py:9)."""
from .standard_gates import (HGate, XGate, IGate, SXGate, SXdgGate, RZGate, PhaseGate, UGate, U2Gate, U3Gate,
                             TGate, TdgGate, CXGate, CCXGate, C3XGate, C4XGate, MCXGate, MCXGrayCode, CPhaseGate)
from .boolean_logic import AND

__all__ = ["HGate", "XGate", "IGate", "SXGate", "SXdgGate", "RZGate", "PhaseGate", "UGate", "U2Gate", "U3Gate",
           "TGate", "TdgGate", "CXGate", "CCXGate", "C3XGate", "C4XGate", "MCXGate", "MCXGrayCode", "CPhaseGate", "AND"]
