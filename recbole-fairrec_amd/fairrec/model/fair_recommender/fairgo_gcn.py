"""FairGo_GCN: same finetune stage as FairGo_PMF (the reference's fairgo_gcn.py differs from fairgo_pmf.py only in the
pretrain stage), whose pretrain model is `torch_geometric.nn.GCN` -- a third-party module the
reference pins nowhere and that is not installed here.  Parity of that stage is UNPINNED, so it is not implemented:
fine-tune from a pretrain checkpoint (`pretrain_model_file_path`) or preloaded weights (`load_pretrain_weight`)."""
from .fairgo_pmf import FairGo_PMF


class FairGo_GCN(FairGo_PMF):
    def calculate_loss(self, interaction, sst_list=None):
        if self.train_stage != 'finetune':
            raise NotImplementedError(
                "FairGo_GCN pretrain uses torch_geometric.nn.GCN (unpinned third party, fairgo_gcn.py:20,52-57,176): "
                "parity unpinned, not on the HIP path; load a pretrain checkpoint and fine-tune")
        return super().calculate_loss(interaction, sst_list)
