"""Graph containers: the callers of the layer contract (reference: models/tf_modules/tf_graphs.py).

The reference's containers build a static TF graph: `add(layer)` dispatches on `type(layer).__name__`, assembles
the input dict (`node_features / data_slice / original_laplacian / lap_slice` + `res_lap`, `block_outputs`,
`inblock_activations`, ...) and calls `layer(input)` once, at construction time.  Here execution is define-by-run:
`add(layer)` records the layer (same name checks and ordering assert), and calling the container on a batch runs
the SAME per-type dispatch, with the same dict keys, the same positional return arities (3 values from `SGC_LL`,
4 from `SGC_LL_Reslap`), the same `stack_laplacians[-batch_size:]` slice, `save_lap` / `save_output` reads and
in-block / out-of-block bookkeeping -- so a layer list written for the reference's containers
(models/networks/basic_AGCN.py:17-49, ResAGCN.py:13-57, DenseAGCN.py:13-85) runs unchanged.

A batch is either a `GraphBatch` (packed rows + CSR Laplacian: the fast path) or the reference's feed as a dict
with the four keys above (what `GraphTopologyMol.batch_to_feed_dict` produces, graph_topology.py:100-135).
"""
from .batch import GraphBatch

_GRAPH_LAYERS = ("MLP", "GraphGatherMol", "GraphPoolMol", "DenseMol")


class SequentialGraphMol(object):
    """tf_graphs.py:12-104.  SequentialGraphMol(n_feat, batch_size=50, max_atom=128)."""

    conv_names = ("SGC_LL",)
    extra_names = ()

    def __init__(self, n_feat, batch_size=50, max_atom=128):
        self.n_feat = n_feat
        self.layers = []
        self.max_atom = max_atom
        self.batch_size = batch_size
        self.output = None
        self.res_L_set = []     # residual Laplacians of every graph of every SGC layer, in layer order
        self.res_W_set = []

    # ---- construction ---------------------------------------------------------------------------------------
    def add(self, layer):
        """Adds a new layer to the model (tf_graphs.py:37-73: names checked here, dispatch repeated per call)."""
        name = type(layer).__name__
        if name in self.conv_names + self.extra_names + _GRAPH_LAYERS:
            if len(self.layers) > 0:
                assert type(self.layers[-1]).__name__ != "GraphGatherMol", \
                    "Cannot use GraphConv or GraphGather layers after a GraphGather"
        self.layers.append(layer)

    # ---- execution ------------------------------------------------------------------------------------------
    def _reset(self):
        self.res_L_set, self.res_W_set = [], []

    def _topology(self, batch):
        """The four topology entries of the reference's input dict for this batch."""
        if isinstance(batch, GraphBatch):
            return {"graph_batch": batch, "data_slice": None, "original_laplacian": None, "lap_slice": None}, batch.X
        missing = [k for k in ("node_features", "data_slice", "original_laplacian", "lap_slice") if k not in batch]
        if missing:
            raise ValueError("feed is missing %s" % missing)
        topo = {k: batch[k] for k in ("data_slice", "original_laplacian", "lap_slice")}
        if "graph_batch" in batch:
            topo["graph_batch"] = batch["graph_batch"]
        return topo, batch["node_features"]

    def _input(self, topo, laplacian=True):
        x = {"node_features": self.output, "data_slice": topo["data_slice"]}
        if laplacian:
            x["original_laplacian"] = topo["original_laplacian"]
            x["lap_slice"] = topo["lap_slice"]
        if topo.get("graph_batch") is not None:
            x["graph_batch"] = topo["graph_batch"]
        return x

    def _adopt_batch(self, topo, out):
        # the first layer fed the reference's padded lists packs them ONCE (GraphBatch.from_padded); later layers
        # reuse that batch through the RowViews it returns
        if topo.get("graph_batch") is None and getattr(out, "batch", None) is not None:
            topo["graph_batch"] = out.batch

    def _run_conv(self, layer, topo, idx):
        self.output, res_L, res_W = layer(self._input(topo))
        self.res_L_set.extend(res_L)
        self.res_W_set.extend(res_W)

    def _run_extra(self, layer, topo, idx):
        raise NotImplementedError

    def __call__(self, batch):
        """Runs every layer on one batch; returns the last layer's output (tf_graphs.py:96-97 return_outputs)."""
        self._reset()
        topo, self.output = self._topology(batch)
        for idx, layer in enumerate(self.layers):
            name = type(layer).__name__
            if name in self.conv_names:
                self._run_conv(layer, topo, idx)
                self._adopt_batch(topo, self.output)
            elif name in self.extra_names:
                self._run_extra(layer, topo, idx)
            elif name == "MLP":
                self.output = layer(self._input(topo, laplacian=False))
            elif name in _GRAPH_LAYERS:                    # GraphGatherMol, GraphPoolMol, DenseMol
                self.output = layer(self._input(topo))
            else:
                self.output = layer(self.output)
        return self.output

    # ---- accessors of the reference (tf_graphs.py:75-104) ------------------------------------------------------
    def get_resL_set(self):
        return self.res_L_set

    def get_resW_set(self):
        return self.res_W_set

    def get_num_output_features(self):
        return self.layers[-1].output_shape[1]

    def return_outputs(self):
        return self.output

    def get_layer(self, layer_id):
        return self.layers[layer_id]

    def parameters(self):
        ps = []
        for layer in self.layers:
            if hasattr(layer, "parameters"):
                ps += list(layer.parameters())
        return ps

    def train(self, mode=True):
        """The learning phase of every layer (the reference's global in_train_phase, model_operatos.py:46-61): dropout is
        active in training only.  `eval()` before predicting."""
        for layer in self.layers:
            if hasattr(layer, "train"):
                layer.train(mode)
            else:
                layer.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    @property
    def uses_learning_phase(self):
        return any(getattr(layer, "uses_learning_phase", False) for layer in self.layers)


class ResidualGraphMol(SequentialGraphMol):
    """tf_graphs.py:107-156: ResNet-style blocks; `BlockEnd` adds the saved output of the previous block."""

    extra_names = ("BlockEnd",)

    def _reset(self):
        super(ResidualGraphMol, self)._reset()
        self.block_outputs = []

    def _first_conv(self, idx):
        return not any(type(l).__name__ in self.conv_names for l in self.layers[:idx])

    def _run_conv(self, layer, topo, idx):
        super(ResidualGraphMol, self)._run_conv(layer, topo, idx)
        if self._first_conv(idx):          # first SGC_LL layer: the initial block residual (tf_graphs.py:126-128)
            self.block_outputs.append(self.output)

    def _run_extra(self, layer, topo, idx):
        x = self._input(topo, laplacian=False)
        x["block_outputs"] = self.block_outputs[-1]
        self.output = layer(x)
        self.block_outputs.append(self.output)


class ResidualGraphMolResLap(ResidualGraphMol):
    """tf_graphs.py:159-220: + the summed Laplacian of earlier `save_lap` layers fed to later ones."""

    conv_names = ("SGC_LL_Reslap",)

    def _reset(self):
        super(ResidualGraphMolResLap, self)._reset()
        self.stack_laplacians = []
        self._last_lap = None

    def _run_conv(self, layer, topo, idx):
        x = self._input(topo)
        # == self.stack_laplacians[-self.batch_size:] (every saved entry is one layer's batch_size Laplacians), but as the
        # packed TileViews the layer returned: no per-graph copies, and the symmetry flag travels with it
        x["res_lap"] = self._last_lap if self._last_lap is not None else []
        self.output, res_L, res_W, L = layer(x)
        self.res_L_set.extend(res_L)
        self.res_W_set.extend(res_W)
        if self._first_conv(idx):
            self.block_outputs.append(self.output)
        if layer.save_lap:                  # a list whose length is the batch size (tf_graphs.py:193-194)
            self.stack_laplacians.extend(L)
            self._last_lap = L

    def get_laplacian(self):
        return self.stack_laplacians


class DenseConnectedGraph(SequentialGraphMol):
    """tf_graphs.py:223-307: DenseNet-style blocks; `DenseBlockEnd` sums every saved in-block activation and
    every earlier block output."""

    extra_names = ("DenseBlockEnd",)

    def __init__(self, n_blocks, *args, **kwargs):
        super(DenseConnectedGraph, self).__init__(*args, **kwargs)
        self.n_blocks = n_blocks

    def _reset(self):
        super(DenseConnectedGraph, self)._reset()
        self.inblock_activations = {b: [] for b in range(self.n_blocks)}
        self.inblock_activations_dim = {b: [] for b in range(self.n_blocks)}
        self.block_outputs = []
        self.block_outputs_dim = []
        self.current_block_id = 0

    def _save_output(self, layer):
        if layer.save_output:               # flat, source-major lists like the reference's `+=` (tf_graphs.py:268-270)
            self.inblock_activations[self.current_block_id] += list(self.output)
            self.inblock_activations_dim[self.current_block_id] += [layer.nb_filter]

    def _run_conv(self, layer, topo, idx):
        super(DenseConnectedGraph, self)._run_conv(layer, topo, idx)
        self._save_output(layer)

    def _run_extra(self, layer, topo, idx):
        assert len(self.inblock_activations[layer.block_id]) > 0
        x = self._input(topo, laplacian=False)
        x["inblock_activations"] = self.inblock_activations[layer.block_id]
        x["inblock_activations_dim"] = self.inblock_activations_dim[layer.block_id]
        x["block_outputs"] = self.block_outputs
        x["block_outputs_dim"] = self.block_outputs_dim
        self.output = layer(x)
        self.block_outputs = self.block_outputs + list(self.output)
        self.block_outputs_dim = self.block_outputs_dim + [layer.output_n_features]
        self.current_block_id += 1


class DenseConnectedGraphResLap(DenseConnectedGraph):
    """tf_graphs.py:310-396: densely connected blocks of `SGC_LL_Reslap` layers."""

    conv_names = ("SGC_LL_Reslap",)

    def _reset(self):
        super(DenseConnectedGraphResLap, self)._reset()
        self.stack_laplacians = []
        self._last_lap = None

    def _run_conv(self, layer, topo, idx):
        x = self._input(topo)
        x["res_lap"] = self._last_lap if self._last_lap is not None else []   # (see ResidualGraphMolResLap._run_conv)
        self.output, res_L, res_W, L = layer(x)
        self.res_L_set.extend(res_L)
        self.res_W_set.extend(res_W)
        self._save_output(layer)
        if layer.save_lap:
            self.stack_laplacians.extend(L)
            self._last_lap = L

    def get_laplacian(self):
        return self.stack_laplacians


class SegmentationGraph(SequentialGraphMol):
    """models/tf_modules/seg_graph.py:9-125: a shared BACKBONE followed by two heads that both start from the backbone's
    output -- a CLASSIFIER head (`add(layer, classifer=True)`, sic) and a SEGMENTATION head (`add(layer, segmentation=True)`).
    The only caller of `GraphPoolMol` in the reference (Seg_AGCN.py:79-81).  Per-type dispatch as there: in the backbone
    `SGC_LL` gets the full topology dict and returns three values, `MLP` gets `node_features` + `data_slice`, everything
    else the full dict; in the classifier head `FCL` gets `node_features` only; in the segmentation head `Merge` is called
    on the raw tensor list (seg_graph.py:67-68).  `FCL` / `MLP` / `Merge` are dispatched by class name like every layer
    (any callable of that name), this package does not ship them (they belong to the trainers).

    Calling the container on a batch returns (classification_outputs, segmentation_outputs)."""

    allowed = ("MLP", "GraphGatherMol", "GraphPoolMol", "SGC_LL", "DenseMol", "FCL", "Merge")

    def __init__(self, n_feat, batch_size=50, num_point=128):
        super(SegmentationGraph, self).__init__(n_feat, batch_size, num_point)
        self.num_point = num_point
        self.backbone_layers, self.classifier_layer, self.segmentation_layer = [], [], []
        self.classifier_output = None
        self.segmentation_output = None

    def add(self, layer, classifer=False, segmentation=False):
        """seg_graph.py:34-109 (the keyword is spelt `classifer` there)."""
        assert type(layer).__name__ in self.allowed
        if classifer:
            assert len(self.backbone_layers) > 0
            self.classifier_layer.append(layer)
        elif segmentation:
            assert len(self.backbone_layers) > 0
            self.segmentation_layer.append(layer)
        else:
            self.backbone_layers.append(layer)
        self.layers = self.backbone_layers + self.classifier_layer + self.segmentation_layer

    def _full(self, topo, feats):
        x = {"node_features": feats, "data_slice": topo["data_slice"], "original_laplacian": topo["original_laplacian"],
             "lap_slice": topo["lap_slice"]}
        if topo.get("graph_batch") is not None:
            x["graph_batch"] = topo["graph_batch"]
        return x

    def __call__(self, batch):
        self._reset()
        topo, self.output = self._topology(batch)
        for layer in self.backbone_layers:                               # seg_graph.py:79-106
            name = type(layer).__name__
            if name == "SGC_LL":
                self.output, res_L, res_W = layer(self._full(topo, self.output))
                self.res_L_set.extend(res_L)
                self.res_W_set.extend(res_W)
                self._adopt_batch(topo, self.output)
            elif name == "MLP":
                x = {"node_features": self.output, "data_slice": topo["data_slice"]}
                if topo.get("graph_batch") is not None:
                    x["graph_batch"] = topo["graph_batch"]
                self.output = layer(x)
            else:
                self.output = layer(self._full(topo, self.output))
        self.classifier_output = self.segmentation_output = self.output
        for layer in self.classifier_layer:                              # seg_graph.py:40-55
            if type(layer).__name__ == "FCL":
                self.classifier_output = layer({"node_features": self.classifier_output})
            else:
                self.classifier_output = layer(self._full(topo, self.classifier_output))
        for layer in self.segmentation_layer:                            # seg_graph.py:57-77
            name = type(layer).__name__
            if name == "SGC_LL":
                self.segmentation_output, res_L, res_W = layer(self._full(topo, self.segmentation_output))
            elif name == "Merge":
                self.segmentation_output = layer(self.segmentation_output)
            else:
                self.segmentation_output = layer(self._full(topo, self.segmentation_output))
        return self.classifier_output, self.segmentation_output

    def get_graph_topology(self):
        raise NotImplementedError("define-by-run containers hold no placeholders: pass the batch to the call")

    def backbone_outputs(self):
        return self.output

    def classification_outputs(self):
        return self.classifier_output

    def segmentation_outputs(self):
        return self.segmentation_output
