function [features, validPts] = getFeaturePoints(input, ImageOriginal)
    %GETFEATUREPOINTS Shadows PP/featureMatching/getFeaturePoints.m: SIFT on the MI355X through aps_mex.
    %   Same signature and outputs (features Kf x 128 single, validPts Kf x 2 double [x y]).  The other detectors of the
    %   reference's switch (getFeaturePoints.m:33-68: vl_SIFT, HARRIS, FAST, SURF, BRISK, ORB, KAZE) are toolbox / VLFeat
    %   calls with no device counterpart: they are forwarded to the reference's own file (aps_call_shadowed), which has
    %   to be on the path below this folder, as INTEGRATION.md sets it up.
    if strcmp(input.detector, 'SIFT') && isa(ImageOriginal, 'uint8')
        [features, validPts] = aps_mex('sift_extract', ImageOriginal, input);
    else
        [features, validPts] = aps_call_shadowed('getFeaturePoints', mfilename('fullpath'), input, ImageOriginal);
    end
end
