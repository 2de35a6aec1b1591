function [features, validPts] = getFeaturePoints(input, ImageOriginal)
    %GETFEATUREPOINTS Shadows PP/featureMatching/getFeaturePoints.m: SIFT on the MI355X through aps_mex.
    %   Same signature and outputs (features Kf x 128 single, validPts Kf x 2 double [x y]); any other
    %   detector falls through to the reference implementation, which must stay on the path below this folder.
    if strcmp(input.detector, 'SIFT') && isa(ImageOriginal, 'uint8')
        [features, validPts] = aps_mex('sift_extract', ImageOriginal, input);
    else
        features = []; validPts = [];
        error('aps:detector', 'detector %s is not built on the device; remove this folder from the path to use the reference', input.detector);
    end
end
