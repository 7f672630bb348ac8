"""Drop-in import path of the reference (`from src.models import CNNLSTM`, notebooks 01/03).

Each module re-exports the MI355X implementation from ``robust_speech_analysis_framework_amd``
under the reference's module and callable names.
"""
