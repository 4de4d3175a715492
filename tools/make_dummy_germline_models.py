"""Small hand-made germline EVS models (the reference tree ships only the somatic and RNA ones: the germline models come with Illumina's
release packages).  Format as L/calibration/VariantScoringModelServer.cpp / RandomForestModel.cpp read it; feature lists as
L/applications/starling/germlineVariantEmpiricalScoringFeatures.hh.  The trees split on the features that the pileup's EVS
accumulators feed (read-position and mapping-quality rank sums), so a wrong accumulator changes GQX / the filters of a record -- and
with --report-evs-features every feature value is printed into the VCF and compared byte for byte.

usage: python tools/make_dummy_germline_models.py OUT_DIR  ->  OUT_DIR/germlineSNVScoringModels.json, germlineIndelScoringModels.json"""
import json
import os
import sys

SNV = ["GenotypeCategory", "SampleRMSMappingQuality", "SiteHomopolymerLength", "SampleStrandBias", "SampleRMSMappingQualityRankSum",
       "SampleReadPosRankSum", "RelativeTotalLocusDepth", "SampleUsedDepthFraction", "ConservativeGenotypeQuality",
       "NormalizedAltHaplotypeCountRatio"]
INDEL = ["GenotypeCategory", "SampleIndelRepeatCount", "SampleIndelRepeatUnitSize", "SampleIndelAlleleBiasLower", "SampleIndelAlleleBias",
         "SampleProxyRMSMappingQuality", "RelativeTotalLocusDepth", "SamplePrimaryAltAlleleDepthFraction", "ConservativeGenotypeQuality",
         "InterruptedHomopolymerLength", "ContextCompressability", "IndelCategory", "NormalizedAltHaplotypeCountRatio",
         "SampleAlleleCountStrandBias"]


def tree(feature, threshold, votes_le, votes_gt):
    """root splits on features[feature] <= threshold -> node 1 else node 2; leaves carry (votes for class 0, votes for class 1)"""
    return {"tree": {"0": [1, 2], "1": [-1, -1], "2": [-1, -1]},
            "node_votes": {"0": [votes_le[0] + votes_gt[0], votes_le[1] + votes_gt[1]], "1": list(votes_le), "2": list(votes_gt)},
            "decisions": {"0": [feature, threshold], "1": [-2, -2.0], "2": [-2, -2.0]}}


def model(features, trees):
    return {"Features": features, "Calibration": {"Scale": 1, "Power": 1}, "FilterCutoff": 3, "ModelType": "RandomForest",
            "Name": "strelka_amd test model", "Version": "1", "Date": "2026-09-26T00:00:00Z", "Model": trees}


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    snv = model(SNV, [tree(SNV.index("SampleReadPosRankSum"), -0.25, (30.0, 70.0), (80.0, 20.0)),
                      tree(SNV.index("SampleRMSMappingQualityRankSum"), 0.1, (60.0, 40.0), (10.0, 90.0)),
                      tree(SNV.index("ConservativeGenotypeQuality"), 30.0, (5.0, 95.0), (90.0, 10.0))])
    indel = model(INDEL, [tree(INDEL.index("ConservativeGenotypeQuality"), 30.0, (5.0, 95.0), (90.0, 10.0)),
                          tree(INDEL.index("SampleIndelAlleleBias"), 1.5, (70.0, 30.0), (20.0, 80.0))])
    for name, key, m in (("germlineSNVScoringModels.json", "SNV", snv), ("germlineIndelScoringModels.json", "INDEL", indel)):
        with open(os.path.join(out, name), "w") as f:
            json.dump({"CalibrationModels": {"Germline": {key: m}}}, f)


if __name__ == "__main__":
    main()
